#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 300 python - <<'PY'
import torch, math, os
from perspectivefields_amd import ops
C = 96
for rows in (128*256, 128*512, 128*768, 128*1024, 128*1536, 204800):
    g = torch.Generator().manual_seed(1)
    d = torch.randn(rows, C, generator=g).cuda(); y = torch.randn(rows, C, generator=g).cuda()
    w1 = torch.randn(4*C, C, generator=g)/math.sqrt(C); b1 = torch.randn(4*C, generator=g)*0.1
    w2 = torch.randn(C, 4*C, generator=g)/math.sqrt(4*C); b2 = torch.randn(C, generator=g)*0.1
    ms = ops.cnx_mlp(d, y, w1, b1, torch.ones(C), torch.zeros(C), 1e-6, w2, b2, torch.ones(C), iters=50)
    print(f"blocks {rows//128:5d}: {ms*1000:.1f} us")
PY
