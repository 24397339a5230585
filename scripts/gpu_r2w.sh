#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest conv / e2e fused upsample"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -k "upsample or conv2d or golden or fused" 2>&1 | tail -3
echo "== bench"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2w_bench.json | cut -c1-160
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2w_layers.txt 2>&1 | head -13
