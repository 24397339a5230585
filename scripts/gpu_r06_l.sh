#!/bin/bash
# r06 call L: after the asm -> MFMA pad (sb_split.h split_f16_mfma_pad): op parity of the four register-direct kernels, e2e suites, and the thin128 A/B with a correct kernel.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_l_pad.log; : > $L
echo "== op tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "thin128 or attn64 or stem7 or attention" 2>&1 | tail -3 | tee -a $L
echo "== e2e"; timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_debug.py tests/test_gpu_fullsize.py tests/test_gpu_r06.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4 | tee -a $L
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
one() { "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  for m in 0 1; do echo -n "PF_THIN128=$m B=32: "; PF_THIN128=$m one timeout 300 $B; done
done 2>&1 | tee -a $L
for i in 1 2; do for m in 0 1; do echo -n "PF_THIN128=$m B=8: "; PF_THIN128=$m one timeout 300 $B --batch 8 --steps 20; done; done 2>&1 | tee -a $L
for i in 1 2; do for m in 0 1; do echo -n "PF_THIN128=$m B=1: "; PF_THIN128=$m one timeout 300 $B --batch 1 --steps 50; done; done 2>&1 | tee -a $L
for m in 0 1; do echo -n "PF_THIN128=$m B=64: "; PF_THIN128=$m one timeout 300 $B --batch 64 --steps 8; done 2>&1 | tee -a $L
