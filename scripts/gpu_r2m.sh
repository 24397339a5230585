#!/bin/bash
# fused LayerNorm (ConvParams::ln) + multi-column dw3x3: GPU suite, bench with / without the fusion, layer table
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 | tee gpurun_out/r2m_test_gpu.log
echo "== bench"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2m_bench.json | cut -c1-160
echo "== bench PF_FUSE_LN=0"; PF_FUSE_LN=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench PF_DW3_VARIANT=4 (old dw3)"; PF_DW3_VARIANT=4 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2m_layers.txt 2>&1 | head -9
grep -E "dwconv3x3|layernorm" gpurun_out/r2m_layers.txt | tail -20
