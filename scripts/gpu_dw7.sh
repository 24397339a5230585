#!/bin/bash
# Packed-fp32 depthwise 7x7 (dw7_pk.hip) on one box: bit-identity tests, isolated sweep (scripts/tune_dw7.py), the class inside the forward (profile_layers) and a
# same-box bench A/B of PF_DW7_VARIANT=4 (scalar kernels) against 7 (packed forms), optionally with PF_DW7_PK_CH / PF_DW7_PK_TH alternatives (DW7_ALTS="16:10 16:20").
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; export TMPDIR=/tmp
BENCH="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
{
echo "==== tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "dwconv7x7" -p no:cacheprovider 2>&1 | tail -8
echo "==== isolated sweep"; TUNE_PK_ONLY=${TUNE_PK_ONLY:-1} timeout 600 python scripts/tune_dw7.py 2>&1 | grep -v amdgpu.ids
echo "==== in the forward"
for V in 4 7; do echo "== PF_DW7_VARIANT=$V"; PF_DW7_VARIANT=$V timeout 300 python scripts/profile_layers.py --out gpurun_out/layers_dw7_$V.txt 2>&1 | grep "dwconv7x7\|total"; done
for A in ${DW7_ALTS:-0:10 0:5}; do echo "== PF_DW7_VARIANT=7 WMAX=40 ch:th $A"; PF_DW7_VARIANT=7 PF_DW7_PK_WMAX=40 PF_DW7_PK_CH=${A%%:*} PF_DW7_PK_TH=${A##*:} timeout 300 python scripts/profile_layers.py --out gpurun_out/layers_dw7_7_$A.txt 2>&1 | grep "dwconv7x7\|total"; done
echo "==== bench A/B"
for rep in 1 2 3; do for V in 4 7; do PF_DW7_VARIANT=$V $BENCH 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dw7 variant', $V, d['value'], d['ms_per_step'])"; done; done
for V in 4 7; do PF_DW7_VARIANT=$V $BENCH --defer-params 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('joined forwards: dw7 variant', $V, d['value'], d['ms_per_step'])"; done
} > $R/gpurun_out/dw7.log 2>&1
tail -70 $R/gpurun_out/dw7.log
