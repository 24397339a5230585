#!/bin/bash
# GPU validation of a change set on ONE box: full -m gpu suite, row-resident GEMM and depthwise 7x7 sweeps, bench with the shipped tile table and with tiles
# autotuned for this library (PF_TUNE_CACHE keeps the tuned table), layer table.  Everything lands under gpurun_out/.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
BENCH="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
echo "== full GPU suite"
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | tail -40 | tee gpurun_out/test_gpu.log | tail -30
echo "== sbr tiles vs LDS tiles"
timeout 300 python scripts/tune_rr.py 2>&1 | tail -60
echo "== depthwise 7x7: streaming (3) vs LDS tile (4)"
timeout 120 python - <<'P'
import torch
from perspectivefields_amd import ops
for (B, H, W, C) in [(32, 20, 20, 384), (32, 10, 10, 768)]:
    mb = 2 * B * H * W * C * 4 / 1e6
    for v, th in [(3, 0), (4, 0), (4, 5), (4, 7), (4, 10)]:
        ms = min(ops.dwconv7x7_bench(v, B, H, W, C, th=th, iters=20) for _ in range(3))
        print(f"dw7 {H}x{W}x{C} B={B} variant {v} th {th}: {ms * 1e3:6.1f} us  {mb / ms / 1e3:6.2f} TB/s")
P
echo "== bench, shipped tile table"; for i in 1 2; do $BENCH 2>&1 | tail -1 | cut -c1-150; done
echo "== bench, tuning library (= product + the experiment of the moment: s_setprio around the halo kernel's MFMAs)"
for i in 1 2; do PF_TUNING_BUILD=1 PF_EXP_SETPRIO=1 $BENCH 2>&1 | tail -1 | cut -c1-150; done
echo "== bench, tiles autotuned for this library (B = 32)"
export PF_TUNE_CACHE=$PWD/gpurun_out/tiles_b32.txt
$BENCH --autotune 1 2>&1 | tail -1 | cut -c1-150
unset PF_TUNE_CACHE
export PF_TILE_TABLE=$PWD/gpurun_out/tiles_b32.txt
for i in 1 2; do $BENCH 2>&1 | tail -1 | cut -c1-150; done
timeout 200 python scripts/profile_layers.py --out gpurun_out/layers_tuned.txt 2>&1 | head -60
grep -c sbr gpurun_out/tiles_b32.txt; grep sbr gpurun_out/tiles_b32.txt | head -40
