#!/bin/bash
# fused MiT Mlp: parity, isolated timing, e2e, bench A/B
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest mit mlp"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "mit_block_mlp" 2>&1 | tail -14
echo "== timing"; timeout 300 python - <<'PY'
import torch, math
from perspectivefields_amd import ops
for C, hs in ((64, 80), (128, 40)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(32, hs, hs, C, generator=g).cuda()
    r = lambda *s: torch.randn(*s, generator=g)
    ms = ops.mit_mlp(x, r(4*C, C)/math.sqrt(C), r(4*C)*0.1, torch.ones(C), torch.zeros(C), 1e-6, r(4*C,1,3,3)*0.3, r(4*C)*0.1, r(C, 4*C)/math.sqrt(4*C), r(C)*0.1, iters=20)
    print(f"mit_mlp C={C} {hs}x{hs} B=32: {ms*1000:.1f} us")
PY
echo "== e2e tests"; timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5
echo "== bench"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2x_bench.json | cut -c1-160
echo "== bench PF_FUSE_MIT_MLP=0"; PF_FUSE_MIT_MLP=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
