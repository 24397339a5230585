#!/bin/bash
# Ablation of the dominant halo tile (tuning build in tree) + clock / power samples while it runs.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; export PF_TUNING_BUILD=1
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power|mclk" | tr '\n' ' '; echo; sleep 0.4; done > gpurun_out/sbh_ablate_smi.txt ) &
SMI=$!
timeout 150 python scripts/tune_sbh_ablate.py 2>&1 | tail -45
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
sort gpurun_out/sbh_ablate_smi.txt | uniq -c | sort -rn | head -8
