#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; export PF_TUNING_BUILD=1
timeout 100 python scripts/tune_sbh_variants.py 2>&1 | tail -45
