#!/bin/bash
# Next round, call 1: the split-f16 scheme with the low activation plane UNSCALED (PF_LO_UNSCALED=1, profiles/r02_mfma_f16_subnormals.md): halves the VALU instructions of the
# halo kernel's K loop (190 -> 103 per chunk, hipcc -S), no wh 2^-11 operand anywhere.  A/B on one box: bench with the product library in the tree, rebuild on the box with
# the switch (~40 s), full GPU suite + bench again.  The results are NOT bit-identical to the product's (other roundings, same accuracy class): the parity margins of the
# golden / oracle tests are what decides.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-140; done
export PF_LO_UNSCALED=1
timeout 300 python -m perspectivefields_amd.build 2>&1 | tail -1
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error" | tail -60 | tee gpurun_out/test_gpu_lo_unscaled.log | tail -8
for i in 1 2; do timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-140; done
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_lo_unscaled.json | cut -c1-200
timeout 120 python scripts/profile_layers.py --out gpurun_out/layers_lo_unscaled.txt 2>&1 | head -12
