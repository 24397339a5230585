#!/bin/bash
# Round-end evidence run: full GPU test suite, smoke, bench (with cpu baseline), other configs, rocprofv3 trace + PMC.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/test_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-3000
echo "== bench noevents"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_noevents.log | cut -c1-200
for P in bf16x3 bf16; do
  echo "== bench $P (reduced precision, not the headline)"; timeout 300 python bench.py --precision $P --steps 10 --warmup 3 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_$P.log | cut -c1-200
done
echo "== configs"; timeout 600 python scripts/bench_configs.py 2>&1 | tail -60
echo "== host-inclusive"; timeout 300 python scripts/bench_e2e_host.py 2>&1 | tail -22
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/layers.txt 2>&1 | head -12
timeout 1200 bash scripts/gpu_rocprof.sh 2>&1 | tail -28
