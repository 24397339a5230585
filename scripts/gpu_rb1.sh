#!/bin/bash
# round 4, call 1: row-block linear layers -- op parity, per-shape timing vs the LDS tiles, end-to-end parity and A/B
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "rb_linear" 2>&1 | tail -30
timeout 300 python scripts/tune_rb.py 2>&1 | tail -40
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_debug.py -x -q 2>&1 | tail -15
for rb in 0 1 0 1; do PF_RB_CHAIN=$rb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb', $rb, d['value'], d['ms_per_step'], d.get('parity'))"; done
PF_RB_CHAIN=1 timeout 300 python scripts/profile_layers.py --batch 32 --out gpurun_out/layers_rb1.txt 2>&1 | tail -75
} > gpurun_out/rb1.log 2>&1
tail -60 gpurun_out/rb1.log
