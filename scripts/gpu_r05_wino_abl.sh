#!/bin/bash
# Winograd 4-wave kernel with one cost removed (tuning build, PF_WINO_ABL=<mask>; results wrong by construction, timing only): which resource bounds the chunk loop?
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PF_TUNING_BUILD=1
for abl in 0 1 2 4 8 16 32 64 84; do
  PF_WINO_ABL=$abl timeout 100 python -c "
from perspectivefields_amd import ops
n = ops.conv_tiles()
best = min(ops.conv2d_bench(32, 80, 80, 256, 256, 3, 1, 1, tile=n.index('wino256x64w4'), iters=5) for _ in range(3))
print('PF_WINO_ABL=$abl  rcu80 (one head)  %.3f ms' % best)" 2>&1 | grep PF_WINO_ABL
done | tee gpurun_out/r05_wino_abl.log
