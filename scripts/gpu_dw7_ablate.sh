#!/bin/bash
# depthwise 7x7 ablation: normal / no stores / no loads, per launch size (in-pipeline timings from the engine profiler)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for d in 0 1 2; do
  echo "== dw7 DIAG $d (0 normal, 1 no stores, 2 no loads)"; PF_DW7_DIAG=$d timeout 300 python scripts/profile_layers.py --out gpurun_out/layers_dw7_diag$d.txt 2>&1 | grep "dwconv7x7"
done
