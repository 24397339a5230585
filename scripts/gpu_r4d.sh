#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 70 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -x -k "golden or oracle or key_order or stream or full_size or split_k" --durations=3 2>&1 | tail -8 | tee gpurun_out/r4d_tests.log
