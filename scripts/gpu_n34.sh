#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== e2e new"; timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -s -k "fields_from or stream or reduced" 2>&1 | grep -v "^$" | tail -22
