#!/bin/bash
# r06 call J: the specialised 7 x 7 image convs (stem7.hip, PF_STEM7): op parity vs fp64 + times, e2e suites (goldens, layer-by-layer taps), same-box A/B.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== op test"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -s -k "stem7x7" 2>&1 | grep -E "stem7x7|passed|failed|FAILED|Error|error" | tail -20 | tee gpurun_out/r06_j_stem7.log
echo "== e2e"; timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_debug.py tests/test_gpu_fullsize.py tests/test_gpu_r06.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
for i in 1 2 3; do
  for m in 0 1; do echo -n "PF_STEM7=$m B=32: "; PF_STEM7=$m timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
done 2>&1 | tee -a gpurun_out/r06_j_stem7.log
for m in 0 1; do echo -n "PF_STEM7=$m B=8: "; PF_STEM7=$m timeout 300 $B --batch 8 --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_j_stem7.log
for m in 0 1; do echo -n "PF_STEM7=$m B=1: "; PF_STEM7=$m timeout 300 $B --batch 1 --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_j_stem7.log
for m in 0 1; do echo -n "PF_STEM7=$m B=64: "; PF_STEM7=$m timeout 300 $B --batch 64 --steps 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_j_stem7.log
