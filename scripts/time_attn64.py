"""Isolated time of the fused stage-1 attention block (attn_block.hip) beside the attention core alone, at the stage-1 shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops
C = 64
g, be = torch.ones(C), torch.zeros(C)
qw, qb, pw, pb = torch.randn(C, C) / 8, torch.zeros(C), torch.randn(C, C) / 8, torch.zeros(C)
for B in (32, 8, 1, 64):
    x = torch.randn(B, 6400, C, device="cuda"); kv = torch.randn(B, 100, 2 * C, device="cuda")
    ms = min(ops.mit_attn64(x, kv, g, be, 1e-6, qw, qb, pw, pb, iters=20)[1] for _ in range(3))
    a = min(ops.sr_attention_variant(x, kv, 1, 1, iters=20)[1] for _ in range(3))
    print(f"B{B} N6400: fused block {1e3*ms:.1f} us | attention core alone {1e3*a:.1f} us")
