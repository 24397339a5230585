#!/bin/bash
# round 4: fused key / value branch -- op parity, isolated timing, e2e A/B (PF_RB_CHAIN 28 vs 60), e2e parity
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
{
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "rb_srkv or rb_linear" 2>&1 | tail -25
timeout 120 python - <<'PY'
import torch, math
from perspectivefields_amd import ops
C=320; B=32
x=torch.randn(B,20,20,C,device="cuda")
g=torch.ones(C); b=torch.zeros(C)
ms=min(ops.rb_srkv(x,g,b,1e-6,torch.randn(C,C,2,2)/36,torch.randn(C),g,b,1e-5,torch.randn(2*C,C)/18,torch.randn(2*C),iters=20) for _ in range(3))
print(f"rb_srkv B=32: {1e3*ms:.1f} us")
PY
for rep in 1 2; do for rb in 28 60; do PF_RB_CHAIN=$rb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb', $rb, d['value'], d['ms_per_step'])"; done; done
PF_RB_CHAIN=60 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb 60 full', d['value'], d['parity'])"
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_debug.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
} > $R/gpurun_out/srkv.log 2>&1
tail -40 $R/gpurun_out/srkv.log
