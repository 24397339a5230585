#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/r2e_test_gpu.log
echo "== bench default (fp32 activations)"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench SBA f16 planes (heuristic tiles)"; PF_SBA=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench SBA f16 planes autotuned"; PF_SBA=1 timeout 600 python bench.py --steps 10 --warmup 3 --autotune 1 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2e_bench_sba.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','parity')})"
echo "== layers default"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2e_layers.txt 2>&1 | head -60
echo "== layers SBA autotuned"; PF_SBA=1 timeout 300 python scripts/profile_layers.py --autotune 1 --out gpurun_out/r2e_layers_sba.txt 2>&1 | head -60
