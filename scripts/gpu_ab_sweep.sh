#!/bin/bash
# Same-box sweep of engine switches (one bench.py run each, no events / extras, alternating, AB_REPS rounds): are the defaults still the fastest forms?
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0 --steps 10 --warmup 3"
for rep in $(seq 1 ${AB_REPS:-2}); do
  for cfg in ${AB_CFGS:-default PF_RB_CHAIN=124 PF_RB_CHAIN=28 PF_RB_CHAIN=0 PF_SIDE_STREAM=0 PF_FUSE_MIT_MLP=0 PF_MIT_MLP_128=1 PF_CNX_MLP_192=1}; do
    if [ "$cfg" = default ]; then v=$($B 2>&1 | tail -1 | grep -o '"value": [0-9.]*'); else v=$(env $cfg $B 2>&1 | tail -1 | grep -o '"value": [0-9.]*'); fi
    echo "$cfg $v"
  done
done | tee gpurun_out/r05_ab_sweep.log
