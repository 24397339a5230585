#!/bin/bash
# Same-box A/B (lib_prev vs lib) of the B = 64 forward that the mixed-resolution workload runs: bench lines and per-shape layer tables.
export TMPDIR=/tmp
mkdir -p gpurun_out
F="timeout 200 python bench.py --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
echo "== B=64 prev"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 $F 2>&1 | tail -1 | cut -c60-130
echo "== B=64 new"; $F 2>&1 | tail -1 | cut -c60-130
PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 timeout 200 python scripts/profile_layers.py --batch 64 --out gpurun_out/layers64_prev.txt 2>&1 | sed -n 2,9p
timeout 200 python scripts/profile_layers.py --batch 64 --out gpurun_out/layers64_new.txt 2>&1 | sed -n 2,9p
