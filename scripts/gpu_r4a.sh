#!/bin/bash
# Full GPU suite at HEAD + the stream-pipeline / cross-stream tests repeated (they compare bit for bit; a scheduling race would be intermittent).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error" | tail -130 | tee gpurun_out/test_gpu.log | tail -4
echo "== stream tests x6"
for i in 1 2 3 4 5 6; do
  timeout 200 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider -k "inference_stream or across_streams or test_key_order" 2>&1 | tail -1
done | tee gpurun_out/test_streams_repeat.log
