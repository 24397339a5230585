#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ops" ; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/test_ops.log
echo "== e2e" ; timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep -v "^\[" | tail -15 | tee gpurun_out/test_e2e.log
echo "== tune" ; timeout 900 python scripts/tune_conv.py 2>&1 | cut -c1-400 | head -12
echo "== layers"; timeout 600 python scripts/profile_layers.py --out gpurun_out/layers.txt 2>&1 | head -30
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-1400
echo "== bench noevents"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_noevents.log | cut -c1-200
