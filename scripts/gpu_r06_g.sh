#!/bin/bash
# r06 call G: full GPU suite at the new defaults (Winograd from 20 x 20 maps on, half-patch geometry, stage-3 split gated), A/B of the Winograd gate.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error" | tail -200 | tee gpurun_out/r06_g_test_gpu.log | tail -6
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
for i in 1 2 3; do
  for m in 40 20; do echo -n "PF_WINO=$m: "; PF_WINO=$m timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
done 2>&1 | tee gpurun_out/r06_g_wino_gate_ab.log
for m in 40 20; do echo -n "PF_WINO=$m B=8 PersNet-size batch: "; PF_WINO=$m timeout 300 $B --batch 8 --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_g_wino_gate_ab.log
for m in 0 96; do echo -n "PF_WINO_MIN_BLOCKS=$m B=1: "; PF_WINO_MIN_BLOCKS=$m timeout 300 $B --batch 1 --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_g_wino_gate_ab.log
