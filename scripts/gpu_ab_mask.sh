#!/bin/bash
# same-box A/B of PF_RB_CHAIN masks (default bench: deferred ParamNet branch on)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
{
for rep in 1 2 3; do for rb in ${MASKS:-0 28 60}; do PF_RB_CHAIN=$rb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb', $rb, d['value'], d['ms_per_step'])"; done; done
} > $R/gpurun_out/ab_mask.log 2>&1
cat $R/gpurun_out/ab_mask.log
