#!/bin/bash
# r06 call O: where the batch gate of the one-kernel Mlp at MiT stage 2 belongs (PF_MIT_MLP_128 = n: from a batch of n images up; 0 = never, 1 = always): B = 2 / 4 / 6, alternating;
# then the touched suites at the new default.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_o_mit_mlp128_gate.log; : > $L
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0"
run() { echo -n "$1 $2: " | tee -a $L; shift; env "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $L; }
for b in 2 4 6; do for rep in 1 2; do for m in 0 1; do run "B=$b" PF_MIT_MLP_128=$m $B --batch $b --steps 60 --warmup 5; done; done; done
echo "== suites at the default"; timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_debug.py tests/test_gpu_fullsize.py tests/test_gpu_r06.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4 | tee -a $L
