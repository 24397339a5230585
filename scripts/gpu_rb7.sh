#!/bin/bash
# round 4: which layers gain from the row-block form in the pipeline (PF_RB_CHAIN bit mask), same box
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
{
for rep in 1 2; do for rb in 0 16 20 28 29 4; do PF_RB_CHAIN=$rb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb', $rb, d['value'], d['ms_per_step'])"; done; done
for rb in 0 29; do echo "== layers rb $rb"; PF_RB_CHAIN=$rb timeout 300 python scripts/profile_layers.py --batch 32 --out gpurun_out/layers_rb$rb.txt 2>&1 | grep "M=   12800 N=\|M=    3200 N=  640\|M=    3200 N=  320 K=  1280\|total"; done
} > $R/gpurun_out/rb7.log 2>&1
tail -40 $R/gpurun_out/rb7.log
