#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1; do echo "== dw7 variant $v"; PF_DW7_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k dwconv7x7 -p no:cacheprovider 2>&1 | tail -2; PF_DW7_VARIANT=$v PF_AUTOTUNE=0 timeout 600 python scripts/profile_layers.py --out gpurun_out/layers_dw7_$v.txt 2>&1 | grep -E "dwconv7x7|total"; done
