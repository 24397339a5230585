"""Times the two spatial-reduction attention kernels (exact fp32 MFMA / split-f16 MFMA) on the MiT-B3 shapes of the B=32 forward."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
out = []
for (N, heads) in ((6400, 1), (1600, 2), (400, 5), (100, 8)):
    C = heads * 64
    q = torch.randn((B, N, C), device="cuda")
    kv = torch.randn((B, 100, 2 * C), device="cuda")
    gf = 4.0 * B * N * C * 100 / 1e9
    r = []
    for variant in (0, 1):
        o, ms = ops.sr_attention_variant(q, kv, heads, variant, iters=20)
        r.append((ms, o))
    d = float((r[0][1] - r[1][1]).abs().max())
    out.append(f"N={N:5d} heads={heads}: fp32 MFMA {1e3*r[0][0]:7.1f} us {gf/r[0][0]:7.1f} GF/ms | split-f16 {1e3*r[1][0]:7.1f} us {gf/r[1][0]:7.1f} GF/ms | speed-up {r[0][0]/r[1][0]:.2f}x | max |diff| {d:.2e}")
txt = "\n".join(out)
open(os.environ.get("TUNE_OUT", "gpurun_out/tune_attn.txt"), "w").write(txt + "\n")
print(txt)
