#!/bin/bash
# Validation of the last commit that touches a source file: whole GPU suite, smoke, the default bench line (what the driver runs).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_validate_head.json | cut -c1-400
