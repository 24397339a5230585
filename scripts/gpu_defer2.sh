#!/bin/bash
# round 4: deferred ParamNet branch -- where to issue it (PF_DEFER_AT) and with which stream priority (PF_DEFER_PRIO)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
{
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 --defer-params 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer 0', d['value'], d['ms_per_step'])"
for at in 0 2 3 4; do for pr in 0 1 -1; do PF_DEFER_AT=$at PF_DEFER_PRIO=$pr timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 --defer-params 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('at', $at, 'prio', $pr, d['value'], d['ms_per_step'])"; done; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 --defer-params 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer 0', d['value'], d['ms_per_step'])"
} > $R/gpurun_out/defer2.log 2>&1
tail -30 $R/gpurun_out/defer2.log
