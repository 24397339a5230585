#!/bin/bash
# fused ConvNeXt MLP: ablation (tuning build)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for D in 0 5 1 3; do
PF_CNX_DIAG=$D timeout 300 python - <<'PY'
import torch, math, os
from perspectivefields_amd import ops
C, rows = 96, 204800
g = torch.Generator().manual_seed(1)
d = torch.randn(rows, C, generator=g).cuda(); y = torch.randn(rows, C, generator=g).cuda()
w1 = torch.randn(4*C, C, generator=g)/math.sqrt(C); b1 = torch.randn(4*C, generator=g)*0.1
w2 = torch.randn(C, 4*C, generator=g)/math.sqrt(4*C); b2 = torch.randn(C, generator=g)*0.1
ms = ops.cnx_mlp(d, y, w1, b1, torch.ones(C), torch.zeros(C), 1e-6, w2, b2, torch.ones(C), iters=20)
print(f"diag {os.environ.get('PF_CNX_DIAG')}: cnx_mlp C={C} rows={rows}: {ms*1000:.1f} us")
PY
done
