#!/bin/bash
# r06 call N: the fused block MLP of MiT stage 2 (mit_mlp_kernel<128>, PF_MIT_MLP_128=1; off since r02) against the three launches, same box, alternating.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_n_mit_mlp128.log; : > $L
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0"
run() { echo -n "$1 $2: " | tee -a $L; shift; env "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $L; }
for rep in 1 2 3 4; do for m in 0 1; do run "B=32" PF_MIT_MLP_128=$m $B --steps 10 --warmup 3; done; done
for rep in 1 2; do for m in 0 1; do run "B=64" PF_MIT_MLP_128=$m $B --batch 64 --steps 8 --warmup 2; done; done
for rep in 1 2; do for m in 0 1; do run "B=8" PF_MIT_MLP_128=$m $B --batch 8 --steps 30 --warmup 5; done; done
for rep in 1 2; do for m in 0 1; do run "B=1" PF_MIT_MLP_128=$m $B --batch 1 --steps 100 --warmup 5; done; done
