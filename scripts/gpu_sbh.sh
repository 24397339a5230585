#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ops conv"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv2d or linear" -p no:cacheprovider -x 2>&1 | tail -3
echo "== tune"; TUNE_ONLY=${TUNE_ONLY:-rcu80,rcu40,rcu20,fold_c1,fold_c2,conv0,conv1} timeout 600 python scripts/tune_conv.py 2>&1 | cut -c1-1100 | sed 's/128x128:.*sb128x128:/... sb128x128:/'
cp gpurun_out/tune_conv.txt gpurun_out/tune_conv_sbh.txt
echo "== bench"; timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/layers_sbh.txt 2>&1 | sed -n 1,16p
