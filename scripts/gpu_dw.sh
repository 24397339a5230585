#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/tune_dw.py 2>&1 | tail -8
for v in 1 2 3; do echo "== dw3 parity variant $v"; PF_DW3_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k dwconv3x3 -p no:cacheprovider 2>&1 | tail -3; done
