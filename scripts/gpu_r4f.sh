#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 22 python scripts/profile_layers.py --out gpurun_out/r4f_layers.txt 2>&1 | head -9
