#!/bin/bash
# wh2 operand made in registers (2 weight planes in LDS): op tests, full tile sweep, bench, layer table
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest ops"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4
echo "== tune all"; TUNE_PREC=0 TUNE_OUT=gpurun_out/r2p_tune_conv.txt timeout 900 python scripts/tune_conv.py 2>&1 | cut -c1-90 | tail -45
echo "== bench"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2p_bench.json | cut -c1-160
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2p_layers.txt 2>&1 | head -9
