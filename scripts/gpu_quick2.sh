#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== resize tests"; timeout 600 python -m pytest tests/test_gpu_resize.py -q -m gpu -p no:cacheprovider 2>&1 | tail -8
echo "== e2e host"; timeout 600 python scripts/bench_e2e_host.py 2>&1 | tail -14
