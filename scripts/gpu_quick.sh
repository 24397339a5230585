#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ops conv"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv2d or linear" -p no:cacheprovider 2>&1 | tail -3
echo "== tune"; TUNE_ONLY=rcu timeout 900 python scripts/tune_conv.py 2>&1 | cut -c1-700 | head -12
echo "== bench noevents"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | cut -c1-200
