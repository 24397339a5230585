#!/bin/bash
# full GPU suite + smoke + a short bench on the working tree
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error|error" | tail -150 | tee gpurun_out/r05_test_gpu.log | tail -25
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench noevents"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-200
