#!/bin/bash
# fused ConvNeXt block MLP: parity + isolated timing + bench with / without
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest cnx mlp"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "convnext_block_mlp" 2>&1 | tail -12
echo "== timing"; timeout 300 python - <<'PY'
import torch, math
from perspectivefields_amd import ops
for C, rows in ((96, 204800), (192, 51200)):
    g = torch.Generator().manual_seed(1)
    d = torch.randn(rows, C, generator=g).cuda(); y = torch.randn(rows, C, generator=g).cuda()
    w1 = torch.randn(4*C, C, generator=g)/math.sqrt(C); b1 = torch.randn(4*C, generator=g)*0.1
    w2 = torch.randn(C, 4*C, generator=g)/math.sqrt(4*C); b2 = torch.randn(C, generator=g)*0.1
    ms = ops.cnx_mlp(d, y, w1, b1, torch.ones(C), torch.zeros(C), 1e-6, w2, b2, torch.ones(C), iters=20)
    fl = 2.0*2.0*rows*C*4*C
    print(f"cnx_mlp C={C} rows={rows}: {ms*1000:.1f} us  {fl/ms/1e9:.1f} TF fp32-equivalent")
PY
echo "== e2e tests"; timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5
echo "== bench"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2q_bench.json | cut -c1-160
echo "== bench PF_FUSE_CNX_MLP=0"; PF_FUSE_CNX_MLP=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== layers"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2q_layers.txt 2>&1 | head -9
grep "KH=1 igemm_sb" gpurun_out/r2q_layers.txt | grep "N=   96\|N=  192\|N=  384 K=    96\|N=  768 K=   192" 
