#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r2i_test_gpu.log
echo "== bench split-K"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench no split-K"; PF_SPLITK=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2i_layers.txt 2>&1 | grep -E "batch|igemm_sb  |M=    3200|other " | head -30
echo "== layers no split"; PF_SPLITK=0 timeout 300 python scripts/profile_layers.py --out gpurun_out/r2i_layers_nosplit.txt 2>&1 | grep -E "batch|igemm_sb  |M=    3200 N=   64|M=    3200 N=  128 K=  2048|M=    3200 N=  320 K=  1280" | head
