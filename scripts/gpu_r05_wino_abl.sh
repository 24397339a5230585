#!/bin/bash
# Winograd 4-wave kernels with one cost removed (tuning build, PF_WINO_ABL=<mask>; results wrong by construction, timing only): which resource bounds the chunk loop?
# masks of wino4d_f2x2_kernel (the LDS form wino256x64w4 and its masks are archived in profiles/r05_rejected/): 1 no LDS reads, 2 no halo staging, 4 no weight requests,
# 8 no transform arithmetic, 16 no barrier, 32 no address updates; 6 = 2 + 4, 63 = MFMAs only, 15 = MFMAs + barrier.  Mask 1 also makes the transform loop invariant)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PF_TUNING_BUILD=1
TILE=${WINO_ABL_TILE:-wino256x64d}
MASKS=${WINO_ABL_MASKS:-"0 1 2 4 8 16 32 7 55 63 15"}
for abl in $MASKS; do
  PF_WINO_ABL=$abl timeout 100 python -c "
from perspectivefields_amd import ops
n = ops.conv_tiles()
best = min(ops.conv2d_bench(32, 80, 80, 256, 256, 3, 1, 1, tile=n.index('$TILE'), iters=5) for _ in range(3))
print('$TILE PF_WINO_ABL=$abl  rcu80 (one head)  %.3f ms' % best)" 2>&1 | grep PF_WINO_ABL
done | tee gpurun_out/r05_winod_abl.log
