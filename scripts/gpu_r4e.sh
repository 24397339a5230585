#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 40 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/r4e_bench.json | cut -c1-250
