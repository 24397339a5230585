#!/bin/bash
# r06 call A: the host-side changes on the GPU (new tests + the suites they touch), a same-box baseline of the bench, the depthwise yardstick.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== new + touched tests"; timeout 1500 python -m pytest tests/test_gpu_r06.py tests/test_gpu_debug.py tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -s -x 2>&1 | grep -E "^\[|passed|failed|FAILED|Error|error" | tail -80 | tee gpurun_out/r06_a_tests.log | tail -40
echo "== bench (short)"; for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r06_a_bench_$i.json | cut -c1-120; done
echo "== dw7 yardstick"; timeout 300 python scripts/dw7_yardstick.py 2>&1 | tail -8
