"""Depthwise 3x3 + GELU: every kernel variant on the four hidden-map shapes of the B=32 forward (pf_op_dwconv3x3_bench, random data).
Variants >= 1000: the multi-column / prefetching kernel, 1000 + 100 block shape + 10 strip height + (columns, prefetch) code (elem.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops
B = int(os.environ.get("TUNE_B", "32"))
SHAPES = {0: (32, 8), 1: (64, 4), 2: (64, 2), 3: (64, 5)}
NP = {0: (1, 1), 1: (2, 0), 2: (2, 1), 3: (2, 2), 4: (1, 2)}
TH = {0: 8, 1: 16, 2: 40}
out = []
for (H, C) in [(80, 256), (40, 512), (20, 1280), (10, 2048)]:
    mb = 2.0 * B * H * H * C * 4 / 1e6
    res = []
    for v in (2, 4):
        ms = ops.dwconv3x3_bench(v, B, H, H, C, iters=20)
        res.append((ms, f"v{v}"))
    for sh, (cqb, xb) in SHAPES.items():
        for np_, (nc, pf) in NP.items():
            if (C // 4) % cqb or H % (xb * nc):
                continue
            for th, rows in TH.items():
                if rows > H and th > 0 and TH[th - 1] >= H:
                    continue
                ms = ops.dwconv3x3_bench(1000 + 100 * sh + 10 * th + np_, B, H, H, C, iters=20)
                res.append((ms, f"q{cqb}x{xb}c{nc}p{pf}t{rows}"))
    best = min(res)
    out.append(f"dw3x3 {H}x{H}x{C} ({mb:.0f} MB): default v4 {res[1][0]*1000:6.1f} us {mb/res[1][0]/1e3:5.2f} TB/s | best {best[1]} {best[0]*1000:6.1f} us {mb/best[0]/1e3:5.2f} TB/s | " +
               " ".join(f"{n}:{ms*1000:.1f}" for ms, n in res))
print("\n".join(out))
os.makedirs("gpurun_out", exist_ok=True)
open(os.environ.get("TUNE_OUT", "gpurun_out/tune_dw.txt"), "w").write("\n".join(out) + "\n")
