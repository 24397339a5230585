#!/bin/bash
# round 4: row-block layers in the pipeline -- op parity, stamps, A/B, per-layer table
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; rm -f $R/gpurun_out/tune_rb.txt
{
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "rb_linear" 2>&1 | tail -3
echo "== stamps"; PF_RB_STAMPS=1 RB_ONLY=1 timeout 120 python scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]"
for rb in 0 1 0 1; do PF_RB_CHAIN=$rb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb', $rb, d['value'], d['ms_per_step'], d['parity']['paramnet_max_abs_delta'], d['parity']['ok'])"; done
PF_RB_CHAIN=1 timeout 300 python scripts/profile_layers.py --batch 32 --out gpurun_out/layers_rb.txt 2>&1 | grep "M=   12800\|M=    3200 N=  640\|M=    3200 N=  320\|attention  \|layernorm  \|dwconv3x3\|total"
} > $R/gpurun_out/rb6.log 2>&1
tail -40 $R/gpurun_out/rb6.log
