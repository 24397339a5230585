#!/bin/bash
# Copies what scripts/gpu_final.sh left in gpurun_out/ into profiles/ under this round's names: collect_evidence.sh r06 <log of the gpurun call>
set -eu
R=${1:?round tag, e.g. r06}; LOG=${2:-}
G=gpurun_out; P=profiles
[ -n "$LOG" ] && cp "$LOG" $P/${R}_final_run.log
cp $G/test_gpu.log $P/${R}_test_gpu.log; cp $G/smoke.log $P/${R}_smoke.log
cp $G/bench.json $P/${R}_bench_final.json
for pair in noevents:noevents noevents2:noevents2 r05_defaults:r05_defaults nowino:r04_path_no_winograd nodefer:joined_forwards fp32_bf16x6:fp32_bf16x6; do
  a=${pair%%:*}; b=${pair##*:}; [ -f $G/bench_$a.json ] && cp $G/bench_$a.json $P/${R}_bench_final_$b.json
done
for f in mixed configs b64_split0 b64_split1; do [ -f $G/bench_$f.json ] && cp $G/bench_$f.json $P/${R}_bench_$f.json; done
cp $G/layers.txt $P/${R}_layers_final.txt
cp $G/rocprof_summary.md $P/${R}_rocprof_summary.md; cp $G/rocprof_summary.json $P/${R}_rocprof_summary.json; cp $G/pmc_traffic.json $P/${R}_pmc_traffic.json
S=$(find $G/rocprof_trace -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $P/${R}_rocprofv3_kernel_stats.csv
python scripts/kernel_resources.py > $P/${R}_kernel_resources.txt 2>/dev/null || true
ls -la $P | grep " ${R}_" | wc -l
