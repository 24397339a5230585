#!/bin/bash
# r06 call M: same-box sweep of the engine's launch-structure switches at the final sources (B = 32, alternating with the default): has an optimum moved with this round's kernels?
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_m_sweep.log; : > $L
B="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
one() { "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo -n "default: " | tee -a $L; one $B | tee -a $L
  for e in PF_MIT_MLP_128=1 PF_CNX_MLP_192=0 PF_SIDE_STREAM=0 PF_SIDE_STREAM=2 PF_RB_CHAIN=28 PF_RB_CHAIN=124 PF_FUSE_LN=0 PF_DEFER_AT=3 PF_WINO_MIN_BLOCKS=0 PF_THIN128=0; do
    echo -n "$e: " | tee -a $L; env $e $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $L
  done
done
echo -n "default: " | tee -a $L; one $B | tee -a $L
