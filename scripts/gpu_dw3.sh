#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
echo "== e2e"; timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -x -s -k "golden or oracle" 2>&1 | grep "img\|passed\|failed" | tail -8
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/layers_dw3.txt 2>&1 | grep -i "total\|dwconv3\|layernorm  \|igemm_sb  "
grep "dwconv3x3" gpurun_out/layers_dw3.txt | tail -4
