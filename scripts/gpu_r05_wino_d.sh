#!/bin/bash
# r05 "wino256x64d" (one MFMA per hand-placed slot): op parity + bit identity with wino256x64c, isolated timing, s_memtime stamps of one block (every slot of chunk 2),
# same-box bench A/B against wino256x64c and the PF_WINO thresholds 40 / 20 / 10.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== op parity"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -s -k "winograd" 2>&1 | grep -E "^\[wino256x64d|passed|failed|FAILED|Error|assert" | head -40 | tee gpurun_out/r05_winod_ops.log
echo "== isolated timing"; timeout 300 python scripts/tune_wino.py 2>&1 | tail -9; cp gpurun_out/tune_wino.txt gpurun_out/r05_winod_tune.txt
echo "== stamps (block 17, one launch of rcu80)"; PF_WINO_STAMPS=1 timeout 120 python -c "
from perspectivefields_amd import ops
n = ops.conv_tiles()
for t in ('wino256x64d', 'wino256x64c'): print(t, ops.conv2d_bench(32, 80, 80, 256, 256, 3, 1, 1, tile=n.index(t), iters=3))" 2>&1 | grep -E "stamps|^wino" | tee gpurun_out/r05_winod_stamps.log | cut -c1-900
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0 --steps 10 --warmup 3"
for rep in 1 2; do
  for cfg in ${WINO_CFGS:-40:wino256x64c 40:wino256x64d}; do
    set -- ${cfg%%:*} ${cfg##*:}
    echo "== bench PF_WINO=$1 PF_WINO_TILE=$2"; PF_WINO=$1 PF_WINO_TILE=$2 $B 2>&1 | tail -1 | cut -c60-100
  done
done 2>&1 | tee gpurun_out/r05_winod_bench_ab.log
