#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_debug.py tests/test_gpu_e2e.py -q -p no:cacheprovider -s -k "debug or float_images or no_reduced or stream or key_order" 2>&1 | grep -E "^\[auto|^\[shadow|passed|failed|FAILED|Error" | tail -30
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do echo "== bench noevents"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c60-100; done
echo "== bench full line"; timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/r05_bench_try.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_try.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d.get("roofline", {}).get("frac"), d.get("e2e_host_stream"), d.get("configs_1"), d.get("cpu_baseline", {}).get("kind"), d.get("parity", {}).get("ok"))
PY
