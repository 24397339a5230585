#!/bin/bash
# r06 call P: the single-weight-buffer form of the stage-2 Mlp kernel (mit_mlp_kernel<128, 8, 8, true>: 63 KB of LDS, two blocks per CU; PF_MIT_MLP_SB=0 = the double-buffered
# form) -- op parity, e2e suites, same-box A/B.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_p_mit_mlp_sb.log; : > $L
echo "== op tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "mlp" 2>&1 | tail -3 | tee -a $L
echo "== e2e"; timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_debug.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3 | tee -a $L
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0"
run() { echo -n "$1 $2: " | tee -a $L; shift; env "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $L; }
for rep in 1 2 3 4; do for m in 0 1; do run "B=32" PF_MIT_MLP_SB=$m $B --steps 10 --warmup 3; done; done
for rep in 1 2; do for m in 0 1; do run "B=64" PF_MIT_MLP_SB=$m $B --batch 64 --steps 8 --warmup 2; done; done
for rep in 1 2; do for m in 0 1; do run "B=8" PF_MIT_MLP_SB=$m $B --batch 8 --steps 30 --warmup 5; done; done
for rep in 1 2; do for m in 0 1; do run "B=4" PF_MIT_MLP_SB=$m $B --batch 4 --steps 60 --warmup 5; done; done
