#!/bin/bash
# First GPU call of the next round (tuning build in tree: PF_TUNING_BUILD=1 python -m perspectivefields_amd.build):
# the DMA weight ring on the 4-wave halo tiles through the WHOLE forward (covers the fused up-sampling / concat instantiations, which the isolated
# conv benchmark cannot reach): golden / oracle tests + bench with the shipped table and with the remapped one, same box.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; export PF_TUNING_BUILD=1
python scripts/remap_tile_table.py gpurun_out/tiles_v2.txt sbh128x64=sbhV2_128x64 sbh128x32=sbhV2_128x32 sbh128x128=sbhV2_128x128
timeout 60 python scripts/tune_sbh_variants.py 2>&1 | tail -40
for T in "" "$PWD/gpurun_out/tiles_v2.txt"; do
  export PF_TILE_TABLE=$T; [ -z "$T" ] && unset PF_TILE_TABLE
  echo "== table: ${T:-shipped}"
  timeout 200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x -k "golden or oracle or batch32 or full_size or fused_upsample" 2>&1 | tail -2
  for i in 1 2; do timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-140; done
done
# linear tiles: ablation + the two scheduling fixes (K-step position carried, loads pinned in front of the MFMAs)
unset PF_TILE_TABLE
timeout 120 python scripts/tune_sb_ablate.py 2>&1 | tail -90
# sub-pixel form of conv_fuse_conv1 (tuning build): op test against the fp64 formula, then the whole forward with PF_SUBPX_CONV1=1 (golden / oracle tests + bench)
timeout 120 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k subpixel -s 2>&1 | tail -8
export PF_SUBPX_CONV1=1
timeout 200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x -k "golden or oracle or batch32" 2>&1 | tail -2
for i in 1 2; do timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-140; done
timeout 120 python scripts/profile_layers.py --out gpurun_out/layers_subpx.txt 2>&1 | head -16
unset PF_SUBPX_CONV1
