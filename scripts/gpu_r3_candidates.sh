#!/bin/bash
# Round 3, GPU call 1: the four changes written at the end of round 2 and never run (VERDICT r02 "Next round" 1) on ONE box, A/B against the product library, plus the
# VALU/MFMA interleave microbenchmark (item 4).  All four library variants are prebuilt in the tree (lib/, lib_lo/, lib_tune/, lib_tune_lo/; see build.py).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
BENCH="timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
two() { for i in 1 2; do $BENCH 2>&1 | tail -1 | cut -c1-150; done; }
echo "== [product] bench"; two
echo "== microbench: VALU instructions per MFMA gap"
timeout 200 scripts/microbench/valu_mfma_interleave 2>&1 | tee gpurun_out/valu_mfma_interleave.md | tail -30
timeout 60 scripts/microbench/valu_mfma_overlap 2>&1 | tee gpurun_out/valu_mfma_overlap.txt | tail -8

echo "== [PF_LO_UNSCALED=1] full GPU suite + bench"
export PF_LO_UNSCALED=1
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error" | tail -60 | tee gpurun_out/test_gpu_lo_unscaled.log | tail -8
two
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_lo_unscaled.json | cut -c1-200
timeout 150 python scripts/profile_layers.py --out gpurun_out/layers_lo_unscaled.txt 2>&1 | head -12
unset PF_LO_UNSCALED

echo "== [PF_TUNING_BUILD=1] DMA ring on the 4-wave halo tiles"
export PF_TUNING_BUILD=1
python scripts/remap_tile_table.py gpurun_out/tiles_v2.txt sbh128x64=sbhV2_128x64 sbh128x32=sbhV2_128x32 sbh128x128=sbhV2_128x128
timeout 100 python scripts/tune_sbh_variants.py 2>&1 | tail -40
for T in "" "$PWD/gpurun_out/tiles_v2.txt"; do
  export PF_TILE_TABLE=$T; [ -z "$T" ] && unset PF_TILE_TABLE
  echo "== table: ${T:-shipped}"
  timeout 250 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x -k "golden or oracle or batch32 or full_size or fused_upsample" 2>&1 | tail -2
  two
done
unset PF_TILE_TABLE
echo "== linear tiles: carried K position / pinned loads"
timeout 200 python scripts/tune_sb_ablate.py 2>&1 | tail -90
echo "== sub-pixel conv_fuse_conv1"
timeout 120 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k subpixel -s 2>&1 | tail -8
export PF_SUBPX_CONV1=1
timeout 250 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x -k "golden or oracle or batch32" 2>&1 | tail -2
two
timeout 150 python scripts/profile_layers.py --out gpurun_out/layers_subpx.txt 2>&1 | head -16
unset PF_SUBPX_CONV1

echo "== [tuning + lo_unscaled] everything together"
export PF_LO_UNSCALED=1 PF_TILE_TABLE=$PWD/gpurun_out/tiles_v2.txt
timeout 250 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x -k "golden or oracle or batch32" 2>&1 | tail -2
two
PF_SUBPX_CONV1=1 $BENCH 2>&1 | tail -1 | cut -c1-150
