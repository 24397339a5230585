"""Isolated time of the specialised 7 x 7 image convs (stem7.hip) at the two shapes of the forward (low-level encoder: stride 2 + ReLU; first patch embedding: stride 4 + LayerNorm)."""
import os, sys, torch, math
sys.path.insert(0, os.getcwd())
from perspectivefields_amd import ops
w = torch.randn(64, 3, 7, 7) / math.sqrt(147); b = torch.zeros(64); g = torch.ones(64); be = torch.zeros(64)
for B in (32, 8):
    x4 = torch.randn(B, 320, 320, 4, device="cuda") * 60; x4[..., 3] = 0
    for stride, ln in ((2, False), (4, True)):
        ms = min(ops.stem7x7(x4, w, b, stride, relu=not ln, ln_gamma=g if ln else None, ln_beta=be if ln else None, iters=20)[1] for _ in range(3))
        print(f"B{B} stride {stride}: stem7 {1e3*ms:.1f} us   (the implicit-GEMM tile inside the B = 32 forward: 185 us at stride 2; 58 us + a 32 us LayerNorm launch at stride 4)")
