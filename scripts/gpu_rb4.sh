#!/bin/bash
# round 4, call 4: timeline stamps of one block + issue-order variants of the step
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; rm -f $R/gpurun_out/tune_rb.txt
{
echo "== stamps"; PF_RB_STAMPS=1 RB_ONLY=1 timeout 120 python scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids
for A in 8 16 1; do echo "== ABL $A"; PF_RB_ABL=$A RB_ONLY=1 timeout 120 python scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids; done
} > $R/gpurun_out/rb4.log 2>&1
tail -60 $R/gpurun_out/rb4.log
