#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r2j_test_gpu.log
echo "== tune conv1/conv0"; TUNE_ONLY=conv TUNE_PREC=0 TUNE_OUT=gpurun_out/r2j_tune_conv.txt timeout 600 python scripts/tune_conv.py 2>&1 | sed 's/.*| auto/auto/' | cut -c1-700 | tail -3
echo "== bench table"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench autotuned"; timeout 600 python bench.py --steps 10 --warmup 3 --autotune 1 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== layers autotuned"; timeout 300 python scripts/profile_layers.py --autotune 1 --out gpurun_out/r2j_layers.txt 2>&1 | grep -E "batch|igemm_sb  |N=   32|N=   64 K=  2880" | head
