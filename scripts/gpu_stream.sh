#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== stream test"; timeout 300 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -x -k "stream" 2>&1 | tail -2
echo "== host bench"; timeout 300 python scripts/bench_e2e_host.py 2>&1 | tail -22
