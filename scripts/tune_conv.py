"""Times every tile configuration of the implicit-GEMM kernel on the conv/GEMM shapes of the B=32 forward
(pf_op_conv2d_bench, random data).  Output: gpurun_out/tune_conv.txt (best tile per shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
# (name, B, H, W, Cin, Cout, K, stride, pad)
SHAPES = [
    ("rcu80", B, 80, 80, 256, 256, 3, 1, 1), ("rcu40", B, 40, 40, 256, 256, 3, 1, 1), ("rcu20", B, 20, 20, 256, 256, 3, 1, 1), ("rcu10", B, 10, 10, 256, 256, 3, 1, 1),
    ("fold_c1", B, 80, 80, 64, 256, 3, 1, 1), ("fold_c2", B, 40, 40, 128, 256, 3, 1, 1), ("fold_c3", B, 20, 20, 320, 256, 3, 1, 1), ("fold_c4", B, 10, 10, 512, 256, 3, 1, 1),
    ("conv0", B, 160, 160, 320, 64, 3, 1, 1), ("conv1", B, 320, 320, 64, 32, 3, 1, 1),
    ("pe1", B, 320, 320, 4, 64, 7, 4, 3), ("ll", B, 320, 320, 4, 64, 7, 2, 3), ("pe2", B, 80, 80, 64, 128, 3, 2, 1), ("pe3", B, 40, 40, 128, 320, 3, 2, 1), ("pe4", B, 20, 20, 320, 512, 3, 2, 1),
    ("s1_qproj", 1, B * 6400, 1, 64, 64, 1, 1, 0), ("s1_fc1", 1, B * 6400, 1, 64, 256, 1, 1, 0), ("s1_fc2", 1, B * 6400, 1, 256, 64, 1, 1, 0), ("s1_sr", B, 80, 80, 64, 64, 8, 8, 0),
    ("s2_qproj", 1, B * 1600, 1, 128, 128, 1, 1, 0), ("s2_fc1", 1, B * 1600, 1, 128, 512, 1, 1, 0), ("s2_fc2", 1, B * 1600, 1, 512, 128, 1, 1, 0), ("s2_sr", B, 40, 40, 128, 128, 4, 4, 0),
    ("s3_qproj", 1, B * 400, 1, 320, 320, 1, 1, 0), ("s3_kv", 1, B * 100, 1, 320, 640, 1, 1, 0), ("s3_fc1", 1, B * 400, 1, 320, 1280, 1, 1, 0), ("s3_fc2", 1, B * 400, 1, 1280, 320, 1, 1, 0), ("s3_sr", B, 20, 20, 320, 320, 2, 2, 0),
    ("s4_qproj", 1, B * 100, 1, 512, 512, 1, 1, 0), ("s4_kv", 1, B * 100, 1, 512, 1024, 1, 1, 0), ("s4_fc1", 1, B * 100, 1, 512, 2048, 1, 1, 0), ("s4_fc2", 1, B * 100, 1, 2048, 512, 1, 1, 0),
    ("cnx_stem", B, 320, 320, 4, 96, 4, 4, 0), ("cnx0_pw1", 1, B * 6400, 1, 96, 384, 1, 1, 0), ("cnx0_pw2", 1, B * 6400, 1, 384, 96, 1, 1, 0),
    ("cnx1_pw1", 1, B * 1600, 1, 192, 768, 1, 1, 0), ("cnx1_pw2", 1, B * 1600, 1, 768, 192, 1, 1, 0), ("cnx2_pw1", 1, B * 400, 1, 384, 1536, 1, 1, 0), ("cnx2_pw2", 1, B * 400, 1, 1536, 384, 1, 1, 0),
    ("cnx3_pw1", 1, B * 100, 1, 768, 3072, 1, 1, 0), ("cnx3_pw2", 1, B * 100, 1, 3072, 768, 1, 1, 0),
]
tiles = ops.conv_tiles()
# TUNE_SB=1: only the split-bf16 tiles, each with fp32 operands ("f") and with split-plane input + output ("p")
SB_ONLY = os.environ.get("TUNE_SB", "0") == "1"
PREC = int(os.environ.get("TUNE_PREC", "0"))  # PF_PRECISION_* of the split tiles: 0 split-f16 (default), 3 exact bf16 split
out = [f"B={B}; precision {PREC}; tiles: " + ", ".join(f"{i}:{t}" for i, t in enumerate(tiles))]
ONLY = [t for t in os.environ.get("TUNE_ONLY", "").split(",") if t]
for name, b, h, w, cin, cout, k, st, pd in SHAPES:
    if ONLY and not any(name.startswith(o) for o in ONLY):
        continue
    ho, wo = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
    flops = 2.0 * b * ho * wo * cout * k * k * cin
    res = []
    iters = 20 if flops < 2e10 else 5
    if SB_ONLY:
        if cin % 32:
            continue
        line = []
        bestf, bestp = (0, ""), (0, "")
        for t in range(len(tiles)):
            if not tiles[t].startswith("sb"):
                continue
            tf = []
            for fmt in (0, 2 if cout % 4 == 0 else 1):
                ms = ops.conv2d_bench(b, h, w, cin, cout, k, st, pd, tile=t, iters=iters, fmt=fmt, precision=3)
                tf.append(flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
            bestf = max(bestf, (tf[0], tiles[t])); bestp = max(bestp, (tf[1], tiles[t]))
            line.append(f"{tiles[t]}:{tf[0]:5.1f}/{tf[1]:5.1f}")
        out.append(f"{name:10s} M={b*ho*wo:8d} N={cout:5d} K={k*k*cin:6d}  fp32-in best {bestf[1]:12s} {bestf[0]:6.1f} TF | planes best {bestp[1]:12s} {bestp[0]:6.1f} TF "
                   f"({(bestp[0]/bestf[0]-1)*100:+5.1f} %) | " + " ".join(line))
        continue
    for t in range(len(tiles)):
        ms = ops.conv2d_bench(b, h, w, cin, cout, k, st, pd, tile=t, iters=iters, precision=PREC)
        if ms <= 0:  # tile not usable for this shape
            continue
        res.append((flops / (ms * 1e-3) / 1e12, t, ms))
    auto_ms = ops.conv2d_bench(b, h, w, cin, cout, k, st, pd, tile=-1, iters=5, precision=PREC)
    best = max(res)
    out.append(f"{name:10s} M={b*ho*wo:8d} N={cout:5d} K={k*k*cin:6d}  best {tiles[best[1]]:10s} {best[0]:6.1f} TF {best[2]:7.3f} ms | auto {flops/(auto_ms*1e-3)/1e12:6.1f} TF | " +
               " ".join(f"{tiles[t]}:{tf:5.1f}" for tf, t, _ in sorted(res, key=lambda r: r[1])))
txt = "\n".join(out)
os.makedirs("gpurun_out", exist_ok=True)
open(os.environ.get("TUNE_OUT", "gpurun_out/tune_conv.txt"), "w").write(txt + "\n")
print(txt)
