#!/bin/bash
# Round-end evidence run: full GPU test suite, smoke, bench (with cpu baseline), other configs, rocprofv3 trace + PMC.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/test_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-2500
echo "== bench noevents"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_noevents.log | cut -c1-200
echo "== configs"; timeout 900 python scripts/bench_configs.py 2>&1 | tail -40
echo "== layers"; timeout 600 python scripts/profile_layers.py --out gpurun_out/layers.txt 2>&1 | head -12
bash scripts/gpu_rocprof.sh 2>&1 | tail -25
