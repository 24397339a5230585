#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest ops+e2e"; timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r2f_test_gpu.log
echo "== tile table"; timeout 1200 python scripts/gen_tile_table.py --out gpurun_out/gfx950_tiles.txt --batches 1,8,32 2>&1 | tail -4
export PF_TILE_TABLE=$PWD/gpurun_out/gfx950_tiles.txt
echo "== bench new table"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2f_layers.txt 2>&1 | grep -E "batch|igemm  |igemm_sb  |K=   147|K=    48|K=   196|K=    64 KH=4" | head
