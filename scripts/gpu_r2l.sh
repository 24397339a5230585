#!/bin/bash
# depthwise 3x3 multi-column / prefetching kernel: parity with the one-column kernel + sweep
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest dwconv3x3"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "dwconv3x3" 2>&1 | tail -4
echo "== tune dw3"; TUNE_OUT=gpurun_out/r2l_tune_dw3.txt timeout 600 python scripts/tune_dw.py 2>&1 | cut -c1-150
