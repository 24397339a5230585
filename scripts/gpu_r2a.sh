#!/bin/bash
# Round 2, first GPU call: parity of the new default scheme (split-f16) + headline-size tests, tile table generation,
# bench of both parity schemes, per-layer tables.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert" | tail -80 | tee gpurun_out/r2a_test_gpu.log
echo "== tile table"; timeout 900 python scripts/gen_tile_table.py --out gpurun_out/gfx950_tiles.txt --batches 1,8,32 2>&1 | tail -12
export PF_TILE_TABLE=$PWD/gpurun_out/gfx950_tiles.txt
echo "== bench fp32 (split-f16)"; timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/r2a_bench.json | cut -c1-4000
echo "== bench fp32_bf16x6"; timeout 300 python bench.py --precision fp32_bf16x6 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/r2a_bench_bf16x6.json | cut -c1-600
echo "== bench heuristic tiles only"; PF_TILE_TABLE= timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2a_bench_heur.json | cut -c1-300
echo "== layers f16"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2a_layers_f16.txt 2>&1 | head -75
echo "== layers bf16x6"; timeout 300 python scripts/profile_layers.py --precision fp32_bf16x6 --out gpurun_out/r2a_layers_bf16x6.txt 2>&1 | head -12
echo "== tune_conv f16 (isolation, all tiles)"; TUNE_PREC=0 TUNE_OUT=gpurun_out/r2a_tune_conv_f16.txt timeout 600 python scripts/tune_conv.py 2>&1 | cut -c1-400 | tail -45
