#!/bin/bash
# Targeted validation of the DMA weight ring in the 16 x 16 / 8-wave halo tile: conv op tests over every tile, partial patches, epilogues, the B=32 oracle test, then a bench.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 80 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x -k "conv2d_all_tiles or conv2d_split_f16_scheme or partial_patches or conv2d_epilogues or batch32" --durations=4 2>&1 | tail -9 | tee gpurun_out/r4c_tests.log
timeout 60 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r4c_bench.json | cut -c1-200
