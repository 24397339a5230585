#!/bin/bash
# r06 call E: attention on the K / V operand image -- bit identity with the staging kernel, per-launch time, e2e goldens, bench A/B (alternating), also B = 8 (configs[1] size).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== attention op tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -s -k "attention" 2>&1 | grep -E "^\[attention image|passed|failed|FAILED|Error|error" | tail -40 | tee gpurun_out/r06_e_attn.log
echo "== e2e goldens etc"; timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_r06.py tests/test_gpu_debug.py -q -m gpu -p no:cacheprovider -k "golden or stage3_batch_split or fused_block_mlps or shadow or row_block or deferred" 2>&1 | tail -3
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
for i in 1 2 3; do
  for m in 0 1; do echo -n "PF_ATTN_IMG=$m B=32: "; PF_ATTN_IMG=$m timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
done 2>&1 | tee -a gpurun_out/r06_e_attn.log
for m in 0 1; do echo -n "PF_ATTN_IMG=$m B=8: "; PF_ATTN_IMG=$m timeout 300 $B --batch 8 --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_e_attn.log
for m in 0 1; do echo -n "PF_ATTN_IMG=$m PF_ATTN_IMG_MIN_B=1 B=1: "; PF_ATTN_IMG_MIN_B=1 PF_ATTN_IMG=$m timeout 300 $B --batch 1 --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_e_attn.log
for m in 0 1; do echo -n "PF_ATTN_IMG=$m B=64: "; PF_ATTN_IMG=$m timeout 300 $B --batch 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_e_attn.log
