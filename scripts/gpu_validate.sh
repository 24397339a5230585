#!/bin/bash
# GPU validation of a change set on ONE box: full -m gpu suite, row-resident GEMM and depthwise 7x7 sweeps, bench with the shipped tile table and with tiles
# autotuned for this library (PF_TUNE_CACHE keeps the tuned table), layer table.  Everything lands under gpurun_out/.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
BENCH="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
echo "== full GPU suite"
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | tail -40 | tee gpurun_out/test_gpu.log | tail -30
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
echo "== tuning library: candidate halo tile 256x128w8 (wave tile 64 x 64) against the shipped tiles"
PF_TUNING_BUILD=1 timeout 200 python - <<'P'
import torch
from perspectivefields_amd import ops
t = ops.conv_tiles()
for name, b, h, c in (("rcu80 x2 heads", 64, 80, 256), ("rcu40 x2 heads", 64, 40, 256), ("rcu20 x2 heads", 64, 20, 256)):
    fl = 2.0 * b * h * h * c * c * 9
    for n in ("sbh256x64w8", "sbh128x64", "sbh128x128", "sbh256x128w8", "sbh256x128w8u"):
        ms = min(ops.conv2d_bench(b, h, h, c, c, 3, 1, 1, tile=t.index(n), iters=5, precision=0) for _ in range(2))
        print(f"{name}: {n:14s} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")
P
echo "== bench, tiles autotuned for this library (B = 32)"
export PF_TUNE_CACHE=$PWD/gpurun_out/tiles_b32.txt
$BENCH --autotune 1 2>&1 | tail -1 | cut -c1-150
unset PF_TUNE_CACHE
export PF_TILE_TABLE=$PWD/gpurun_out/tiles_b32.txt
for i in 1 2; do $BENCH 2>&1 | tail -1 | cut -c1-150; done
timeout 200 python scripts/profile_layers.py --out gpurun_out/layers_tuned.txt 2>&1 | head -60
