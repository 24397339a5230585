#!/bin/bash
# round 4, call 3: stream kernel with L stages of read-ahead, epilogue constants from LDS
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; rm -f $R/gpurun_out/tune_rb.txt
{
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "rb_linear or sr_attention" 2>&1 | tail -5
for L in 2 4; do echo "== LEAD $L"; PF_RB_LEAD=$L RB_ONLY=1 timeout 120 python scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids; done
for A in 3 7; do echo "== ABL $A"; PF_RB_ABL=$A RB_ONLY=1 timeout 120 python scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids; done
} > $R/gpurun_out/rb3.log 2>&1
tail -40 $R/gpurun_out/rb3.log
