#!/bin/bash
# r06 call K: the thin 128 -> 128 projections of MiT stage 2 (thin_linear.hip, PF_THIN128): op parity vs fp64 + times, e2e suites (goldens, layer-by-layer taps), same-box A/B.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== op test"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -s -k "thin128" 2>&1 | grep -E "thin128|passed|failed|FAILED|Error|error" | tail -20 | tee gpurun_out/r06_k_thin128.log
echo "== e2e"; timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_debug.py tests/test_gpu_fullsize.py tests/test_gpu_r06.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
for i in 1 2 3; do
  for m in 0 1; do echo -n "PF_THIN128=$m B=32: "; PF_THIN128=$m timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
done 2>&1 | tee -a gpurun_out/r06_k_thin128.log
for m in 0 1; do echo -n "PF_THIN128=$m B=8: "; PF_THIN128=$m timeout 300 $B --batch 8 --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_k_thin128.log
for m in 0 1; do echo -n "PF_THIN128=$m B=1: "; PF_THIN128=$m timeout 300 $B --batch 1 --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_k_thin128.log
for m in 0 1; do echo -n "PF_THIN128=$m B=64: "; PF_THIN128=$m timeout 300 $B --batch 64 --steps 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_k_thin128.log
