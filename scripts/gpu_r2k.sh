#!/bin/bash
# HEAD check after the split-K and sbh256x32 commits: GPU suite, conv0/conv1 tile sweep, bench (table / no split-K / autotuned), layer tables
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r2k_test_gpu.log
echo "== tune conv1/conv0"; TUNE_ONLY=conv TUNE_PREC=0 TUNE_OUT=gpurun_out/r2k_tune_conv.txt timeout 300 python scripts/tune_conv.py 2>&1 | sed 's/.*| auto/auto/' | cut -c1-700 | tail -3
echo "== bench table"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2k_bench.json | cut -c1-160
echo "== bench no split-K"; PF_SPLITK=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench autotuned"; timeout 400 python bench.py --steps 10 --warmup 3 --autotune 1 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2k_layers.txt 2>&1 | head -12
echo "== layers autotuned"; timeout 300 python scripts/profile_layers.py --autotune 1 --out gpurun_out/r2k_layers_auto.txt 2>&1 | grep -E "batch|igemm_sb  |N=   32|N=   64 K=  2880" | head
