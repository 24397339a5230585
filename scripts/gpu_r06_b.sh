#!/bin/bash
# r06 call B: new + touched GPU tests, the default bench line (new keys: step_split_ms, default_auto, configs_4), same-box short benches.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== new + touched tests"; timeout 1800 python -m pytest tests/test_gpu_r06.py tests/test_gpu_debug.py tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|FAILED|Error|error" | tail -120 | tee gpurun_out/r06_b_tests.log | tail -60
echo "== bench default"; timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/r06_b_bench.json | cut -c1-200
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_b_bench.json"))
for k in ("value", "step_split_ms", "default_auto", "configs_4", "configs_1", "e2e_host_stream", "latency_ms", "roofline_depthwise_all", "parity"):
    print(k, json.dumps(d.get(k))[:400])
PY
