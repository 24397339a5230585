#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
echo "== e2e"; timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -x -s -k "reduced or golden or plane" 2>&1 | grep -v "^$" | tail -14
for P in fp32 bf16x3 bf16; do
  echo "== bench $P"; PF_PRECISION=$P timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
done
echo "== layers bf16"; PF_PRECISION=bf16 timeout 300 python scripts/profile_layers.py --out gpurun_out/layers_bf16.txt 2>&1 | head -14
