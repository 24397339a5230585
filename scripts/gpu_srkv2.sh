#!/bin/bash
# round 4: L2 warm-up of the weight streams -- op parity, in-pipeline layer times, e2e A/B
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
{
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "rb_srkv or rb_linear" 2>&1 | tail -3
for rep in 1 2; do for rb in 0 28 60; do PF_RB_CHAIN=$rb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb', $rb, d['value'], d['ms_per_step'])"; done; done
for rb in 0 60; do echo "== layers rb $rb"; PF_RB_CHAIN=$rb timeout 300 python scripts/profile_layers.py --batch 32 --out gpurun_out/layers_rb$rb.txt 2>&1 | grep "M=   12800 N=\|M=    3200 N=  640\|M=    3200 N=  320 K=  1280\|M=    3200 N=  960\|total\|layernorm  \|attention  "; done
} > $R/gpurun_out/srkv2.log 2>&1
tail -40 $R/gpurun_out/srkv2.log
