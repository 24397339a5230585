#!/bin/bash
# tile table with the fused-LayerNorm launch forms; GPU suite; bench with the new table
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tile table"; timeout 1500 python scripts/gen_tile_table.py --out gpurun_out/gfx950_tiles.txt --batches 1,8,32,64 2>&1 | tail -4
export PF_TILE_TABLE=$PWD/gpurun_out/gfx950_tiles.txt
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r2n_test_gpu.log
echo "== bench"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2n_bench.json | cut -c1-160
echo "== bench PF_FUSE_LN=0"; PF_FUSE_LN=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2n_layers.txt 2>&1 | head -9
