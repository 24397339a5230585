"""Row-block linear layers (rb_gemm.hip) against the best LDS tile on the MiT stage-3 shapes of the B = 32 forward (random data, isolated launches).
RB_ONLY=1: only the rb launches (PMC passes); PF_RB_ABL=<mask>: timing-only ablation forms (1 no weight refills, 2 no MFMAs, 4 no A fragment reads)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
ONLY = os.environ.get("RB_ONLY", "0") == "1"
SHAPES = [("s3_q/proj", 400, 320, 320, False), ("s3_q+LN", 400, 320, 320, True), ("s3_kv+LN", 100, 320, 640, True), ("s3_fc1+LN", 400, 320, 1280, True), ("s3_fc2", 400, 1280, 320, False)]
if os.environ.get("PF_RB_ABL"):
    SHAPES = [s for s in SHAPES if not s[4]]  # the ablation forms exist without LayerNorm only
    SHAPES.append(("s3_fc1", 400, 320, 1280, False))
tiles = ops.conv_tiles()
out = []
for name, tokens, K, N, ln in SHAPES:
    rows = B * tokens
    x = torch.randn(rows, K, device="cuda")
    w, b = torch.randn(N, K) / K ** 0.5, torch.randn(N)
    g, be = (torch.ones(K), torch.zeros(K)) if ln else (None, None)
    ms = min(ops.rb_linear(x, w, b, tokens, gamma=g, beta=be, iters=20) for _ in range(3))
    flops = 2.0 * rows * K * N
    line = f"{name:10s} M={rows:6d} K={K:5d} N={N:5d}  rb {1e3*ms:7.1f} us {flops/(ms*1e-3)/1e12:6.1f} TF"
    if not ONLY:
        best = (1e9, "")
        for t in range(len(tiles)):
            if not tiles[t].startswith("sb") or tiles[t].startswith("sbh"):
                continue
            m = ops.conv2d_bench(1, rows, 1, K, N, 1, 1, 0, tile=t, iters=20, precision=0)
            if m > 0:
                best = min(best, (m, tiles[t]))
        line += f" | best LDS tile {best[1]:12s} {1e3*best[0]:7.1f} us {flops/(best[0]*1e-3)/1e12:6.1f} TF"
    out.append(line)
txt = "\n".join(out)
os.makedirs("gpurun_out", exist_ok=True)
open(os.environ.get("TUNE_OUT", "gpurun_out/tune_rb.txt"), "a").write(f"ABL={os.environ.get('PF_RB_ABL', '0')}\n" + txt + "\n")
print(txt)
