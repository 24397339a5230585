#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ops"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -x -k "dwconv7 or upsample or plane" 2>&1 | tail -3
for v in "1 0" "1 40" "0 40"; do set -- $v
  echo "== upsample variant $1 dw7 TH $2"; PF_UPSAMPLE_VARIANT=$1 PF_DW7_TH=$2 PF_SBA=0 timeout 600 python scripts/profile_layers.py --out gpurun_out/layers_elem_$1_$2.txt 2>&1 | grep -i "dwconv7\|total\|upsample"
done
