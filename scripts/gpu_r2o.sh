#!/bin/bash
# whole-N 256x256 halo tiles: correctness of every tile on the 3x3 cases + sweep on the decoder shapes
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest conv tiles"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "conv2d" 2>&1 | tail -4
echo "== tune rcu/fold"; TUNE_ONLY=rcu,fold TUNE_PREC=0 TUNE_OUT=gpurun_out/r2o_tune_conv.txt timeout 600 python scripts/tune_conv.py 2>&1 | sed 's/128x128: .*sb128x128f2:[ 0-9.]*//' | cut -c1-400
