#!/bin/bash
# round 4: deferred ParamNet branch -- A/B on one box + parity of the e2e suite
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
{
for rep in 1 2; do for d in 0 1; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 --defer-params $d 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer', $d, d['value'], d['ms_per_step'])"; done; done
for d in 0 1; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --defer-params $d 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer', $d, 'full line', d['value'], d['ms_per_step'], d['parity'])"; done
for d in 0 1; do timeout 300 python bench.py --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --events-in-timed 0 --defer-params $d 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B64 defer', $d, d['value'], d['ms_per_step'])"; done
for d in 0 1; do timeout 300 python bench.py --batch 8 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 --defer-params $d 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B8 defer', $d, d['value'], d['ms_per_step'])"; done
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -4
} > $R/gpurun_out/defer.log 2>&1
tail -30 $R/gpurun_out/defer.log
