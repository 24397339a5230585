#!/bin/bash
# depthwise 3x3 variants IN THE PIPELINE (per-size rows of the layer table); product build: only 1223 / 1100 / 1314 exist -> tuning build
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for V in -1 1100 1102 1104 1112 1122 1202 1223 1002 1012; do
  if [ "$V" = "-1" ]; then unset PF_DW3_VARIANT; else export PF_DW3_VARIANT=$V; fi
  echo "== PF_DW3_VARIANT=$V"; timeout 120 python scripts/profile_layers.py --out gpurun_out/r2z_layers_$V.txt > /dev/null 2>&1; grep "dwconv3x3_gelu  " gpurun_out/r2z_layers_$V.txt | tail -4
done
