#!/bin/bash
# pk_opsel_partners.py beside the forward under different engine switches (one process each): which part of the forward makes the src1-high-half forms fail?
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
P="Paramnet-360Cities-edina-centered"
run() { env PK_FORWARD="$1" PK_LABEL="$2" ${3:-} timeout 120 python scripts/microbench/pk_opsel_partners.py 2>&1 | grep "^beside"; }
run $P "default switches"
run $P "PF_SIDE_STREAM=0" "PF_SIDE_STREAM=0"
run $P "PF_RB_CHAIN=0" "PF_RB_CHAIN=0"
run $P "PF_RB_CHAIN=0 PF_SIDE_STREAM=0 PF_FUSE_CNX_MLP=0 PF_FUSE_MIT_MLP=0" "PF_RB_CHAIN=0 PF_SIDE_STREAM=0 PF_FUSE_CNX_MLP=0 PF_FUSE_MIT_MLP=0"
run $P "PF_DW7_VARIANT=4" "PF_DW7_VARIANT=4"
run "PersNet-360Cities" "default switches (no ParamNet)"
