#!/bin/bash
# Round-end evidence run on ONE box at the final HEAD: full GPU test suite, smoke, bench (default line with cpu baseline + extras; no-events; other precisions;
# the mixed-resolution workload; other BASELINE configs), per-layer table, rocprofv3 kernel trace + PMC passes.  Everything lands in gpurun_out/; what is to be judged
# is copied into profiles/ (r03_*).  The tile table is the shipped one (what the driver's run uses).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error" | tail -150 | tee gpurun_out/test_gpu.log | tail -5
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/bench.json | cut -c1-300
echo "== bench noevents"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_noevents.json | cut -c1-160
echo "== bench joined forwards (no deferred ParamNet branch)"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 --defer-params 0 2>&1 | tail -1 | tee gpurun_out/bench_nodefer.json | cut -c1-160
echo "== bench LDS tiles only (PF_RB_CHAIN=0)"; PF_RB_CHAIN=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_norb.json | cut -c1-160
echo "== bench mixed (configs[4])"; timeout 300 python bench.py --workload mixed --batch 64 --steps 8 --warmup 2 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_mixed.json | cut -c1-400
for P in fp32_bf16x6 bf16; do
  echo "== bench $P"; timeout 300 python bench.py --precision $P --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_$P.json | cut -c1-160
done
echo "== configs"; timeout 600 python scripts/bench_configs.py 2>&1 | grep -E "config|images_per_sec|agreement" | head -30
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/layers.txt 2>&1 | head -12
timeout 1500 bash scripts/gpu_rocprof.sh 2>&1 | tail -45
