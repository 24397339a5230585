#!/bin/bash
# Round 2, third GPU call: full suite (graph replay, two-stream mode, attention scale), bench default / two streams, latency.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r2c_test_gpu.log
echo "== bench default"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r2c_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','latency_ms','with_device_resize','parity')}); print(d.get('roofline_dwconv7x7')); print(d.get('attention'))"
echo "== bench no events"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench 2 streams no events"; PF_STREAMS=2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench no graph latency"; PF_GRAPH_MAX_BATCH=0 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d.get('latency_ms'))"
