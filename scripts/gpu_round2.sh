#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== ops" ; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider 2>&1 | tail -30 | tee gpurun_out/test_ops.log
echo "== e2e" ; timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/test_e2e.log
echo "== tune" ; timeout 900 python scripts/tune_conv.py 2>&1 | tail -60
echo "== layers"; timeout 600 python scripts/profile_layers.py --out gpurun_out/layers.txt 2>&1 | tail -70
echo "== bench"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench.log

echo "== bench noevents"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_noevents.log
