#!/bin/bash
# Short end-of-round check: full GPU suite, smoke, default bench (+ cpu baseline), bench without events, layer table, LDS PMC pass.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/test_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/smoke.log
echo "== bench"; timeout 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-200
echo "== bench noevents"; timeout 200 python bench.py --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_noevents.log | cut -c1-200
echo "== layers"; timeout 200 python scripts/profile_layers.py --out gpurun_out/layers.txt 2>&1 | head -14
timeout 200 bash scripts/gpu_pmc_lds.sh 2>&1 | tail -12
