#!/bin/bash
# r05 Winograd prototype: op parity vs fp64, isolated timing against the halo tiles (+ the s_memtime stamps of one block: PF_WINO_STAMPS=1),
# optionally (WINO_E2E=1) the end-to-end goldens with the Winograd convs on and a bench A/B of PF_WINO = 0 / 80 / 40.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== op parity"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -s -k "winograd" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | head -40 | tee gpurun_out/r05_wino_ops.log
echo "== isolated timing"; timeout 300 python scripts/tune_wino.py 2>&1 | tail -9
echo "== stamps (block 17, one launch of rcu80)"; PF_WINO_STAMPS=1 timeout 120 python -c "
from perspectivefields_amd import ops
n = ops.conv_tiles()
for t in ('wino256x64d', 'wino256x64c'): print(t, ops.conv2d_bench(32, 80, 80, 256, 256, 3, 1, 1, tile=n.index(t), iters=3))" 2>&1 | grep -E "stamps|^wino" | tee gpurun_out/r05_wino_stamps.log | cut -c1-1500
if [ -n "${WINO_E2E:-}" ]; then
echo "== e2e goldens (PF_WINO default)"; timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_debug.py -q -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error" | tail -40 | tee gpurun_out/r05_wino_e2e.log | tail -12
fi
if [ -n "${WINO_BENCH:-}" ]; then
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0 --steps 10 --warmup 3"
for rep in 1 2; do
  for cfg in "0 wino256x64d" "40 wino256x64c" "40 wino256x64d" "20 wino256x64d"; do
    set -- $cfg
    echo "== bench PF_WINO=$1 PF_WINO_TILE=$2"; PF_WINO=$1 PF_WINO_TILE=$2 $B 2>&1 | tail -1 | cut -c60-100
  done
done 2>&1 | tee gpurun_out/r05_wino_bench_ab.log
fi
