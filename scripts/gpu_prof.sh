#!/bin/bash
# Per-layer event profile + rocprofv3 kernel trace of the bench command.  Outputs in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== layers"; timeout 600 python scripts/profile_layers.py --out gpurun_out/layers.txt 2>&1 | tail -70
echo "== bench (no events)"; timeout 600 python bench.py --steps 5 --warmup 2 --events-in-timed 0 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_noevents.log
echo "== bench"; timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench.log
echo "== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_run.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/rocprof -name "*stats*" | head; f=$(find gpurun_out/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep only the small stats csv (the trace itself can be large)
find gpurun_out/rocprof -name "*kernel_trace.csv" -size +20M -delete
