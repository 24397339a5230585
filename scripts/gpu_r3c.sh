#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest planes"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "split_f16_plane_operands" 2>&1 | tail -4
echo "== e2e SBA_HEADS"; timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -k "PF_SBA_HEADS" -s 2>&1 | grep -E "^\[|passed|failed" | tail -3
for i in 1 2; do
for M in 0 1; do
echo -n "PF_SBA_HEADS=$M: "; PF_SBA_HEADS=$M timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c56-120
done; done
PF_SBA_HEADS=1 timeout 200 python scripts/profile_layers.py --out gpurun_out/r3c_layers.txt 2>&1 | sed -n 8,14p
