#!/bin/bash
# r06 call C: the stage-3 batch split (PF_S3_SPLIT) -- bit identity, then same-box A/B of the bench (alternating, 3 x), also at B = 64 and with joined forwards.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== split test"; timeout 600 python -m pytest tests/test_gpu_r06.py -q -m gpu -p no:cacheprovider -s -k "stage3_batch_split" 2>&1 | grep -E "^\[|passed|failed|FAILED|Error|error" | tail -10
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
for i in 1 2 3; do
  for m in 0 1; do echo -n "PF_S3_SPLIT=$m B=32: "; PF_S3_SPLIT=$m timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
done 2>&1 | tee gpurun_out/r06_c_split_ab.log
for m in 0 1; do echo -n "PF_S3_SPLIT=$m B=64: "; PF_S3_SPLIT=$m timeout 300 $B --batch 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_c_split_ab.log
for m in 0 1; do echo -n "PF_S3_SPLIT=$m B=32 joined: "; PF_S3_SPLIT=$m timeout 300 $B --defer-params 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r06_c_split_ab.log
