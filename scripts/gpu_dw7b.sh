#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ops"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -x -k "dwconv7" 2>&1 | tail -1
echo "== ops cpb"; PF_DW7_CPB=1 timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -x -k "dwconv7" 2>&1 | tail -1
for v in "0 0" "1 0" "1 80" "1 20"; do set -- $v
  echo "== dw7 CPB $1 TH $2"; PF_DW7_CPB=$1 PF_DW7_TH=$2 PF_SBA=0 timeout 600 python scripts/profile_layers.py --out gpurun_out/layers_dw7_$1_$2.txt 2>&1 | grep -i "dwconv7"
done
