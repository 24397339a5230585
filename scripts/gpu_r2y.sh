#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do
for M in 0 1 2; do
echo -n "PF_SIDE_STREAM=$M: "; PF_SIDE_STREAM=$M timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c56-120
done; done
