#!/bin/bash
# Round-end evidence run (r06) on ONE box at the final HEAD: full GPU test suite, smoke, bench (default line with cpu baseline + extras; no-events; the r04 path without
# Winograd; joined forwards; the exact mode; the mixed-resolution workload; configs[1] / [4]), per-layer table, rocprofv3 kernel trace + PMC passes.  Everything lands in
# gpurun_out/; what is to be judged is copied into profiles/ (r06_*).  The tile table is the shipped one (what the driver's run uses).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error" | tail -170 | tee gpurun_out/test_gpu.log | tail -5
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/bench.json | cut -c1-300
echo "== bench noevents"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_noevents.json | cut -c1-160
echo "== bench r05 defaults (PF_WINO=40 PF_WINO_HALF=0 PF_S3_SPLIT=0 PF_ATTN64=0 PF_STEM7=0 PF_THIN128=0 PF_MIT_MLP_128=0: square Winograd patches from 40 x 40 maps on, stage 1 with separate q / attention / proj launches)"; PF_WINO=40 PF_WINO_HALF=0 PF_S3_SPLIT=0 PF_ATTN64=0 PF_STEM7=0 PF_THIN128=0 PF_MIT_MLP_128=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_r05_defaults.json | cut -c1-160
echo "== bench r04 path (PF_WINO=0: direct halo tiles for every 3x3 conv)"; PF_WINO=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_nowino.json | cut -c1-160
echo "== bench batch 64 (stage-3 split on / off)"; for m in 1 0; do PF_S3_SPLIT=$m timeout 300 python bench.py --batch 64 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_b64_split$m.json | cut -c1-160; done
echo "== bench noevents (again)"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_noevents2.json | cut -c1-160
echo "== bench joined forwards (no deferred ParamNet branch)"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 --defer-params 0 2>&1 | tail -1 | tee gpurun_out/bench_nodefer.json | cut -c1-160
echo "== bench mixed (configs[4])"; timeout 300 python bench.py --workload mixed --batch 64 --steps 8 --warmup 2 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_mixed.json | cut -c1-400
echo "== bench fp32_bf16x6"; timeout 300 python bench.py --precision fp32_bf16x6 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/bench_fp32_bf16x6.json | cut -c1-160
echo "== configs"; timeout 600 python scripts/bench_configs.py 2>&1 | grep -E "config|images_per_sec|agreement" | head -30
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/layers.txt 2>&1 | head -12
timeout 1500 bash scripts/gpu_rocprof.sh 2>&1 | tail -50
