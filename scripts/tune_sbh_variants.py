"""DMA-ring variants of the halo tiles ("sbhV2_*" two LDS weight buffers, "sbhV3_*" three; right results) against the shipped tiles -- TUNING BUILD only.
Bit check of every variant against its shipped tile (incl. a concat case), then ms / TF per (shape, tile).  Output: gpurun_out/sbh_variants.txt"""
import os, sys
import torch  # before the library: one HIP runtime per process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
tiles = ops.conv_tiles()
PAIRS = {"sbhV2_128x64": "sbh128x64", "sbhV3_128x64": "sbh128x64", "sbhV2_128x32": "sbh128x32", "sbhV2_256x32": "sbh256x32", "sbhV2_128x128": "sbh128x128",
         "sbhV2_256x64w8": "sbhREG", "sbh256x64w8": "sbhREG"}
PLAN = [
    ("rcu80", B, 80, 80, 256, 256, ["sbhREG", "sbh256x64w8", "sbhV2_256x64w8", "sbh128x64", "sbhV2_128x64", "sbhV3_128x64"]),
    ("rcu40", B, 40, 40, 256, 256, ["sbhREG", "sbh256x64w8", "sbhV2_256x64w8", "sbh128x64", "sbhV2_128x64", "sbhV3_128x64", "sbh128x128", "sbhV2_128x128"]),
    ("conv0", B, 160, 160, 320, 64, ["sbhREG", "sbh256x64w8", "sbh128x64", "sbhV2_128x64", "sbhV3_128x64"]),
    ("conv1", B, 320, 320, 64, 32, ["sbh128x32", "sbhV2_128x32", "sbh256x32", "sbhV2_256x32"]),
]
out = []
torch.manual_seed(0)
for v, base in PAIRS.items():
    if v not in tiles or base not in tiles:
        out.append(f"bit check {v}: tile missing"); continue
    ok = True
    for (b, h, w, c1, c2, cout) in [(2, 40, 40, 64, 0, 128), (1, 33, 47, 32, 0, 96), (1, 16, 16, 96, 0, 64), (1, 24, 40, 32, 32, 64)]:
        x = torch.randn(b, h, w, c1, device="cuda")
        x2 = torch.randn(b, h, w, c2, device="cuda") if c2 else None
        wt = torch.randn(cout, c1 + c2, 3, 3, device="cuda") * 0.05
        bias = torch.randn(cout, device="cuda")
        y0 = ops.conv2d(x, wt, bias, stride=1, pad=1, tile=tiles.index(base), x2=x2)
        y1 = ops.conv2d(x, wt, bias, stride=1, pad=1, tile=tiles.index(v), x2=x2)
        ok = ok and bool(torch.equal(y0, y1))
    out.append(f"bit check {v} vs {base}: {'identical' if ok else 'DIFFERS'}")
for name, b, h, w, cin, cout, want in PLAN:
    flops = 2.0 * b * h * w * cout * 9 * cin
    res = {n: [] for n in want if n in tiles}
    for rep in range(2):
        for n in res:
            res[n].append(ops.conv2d_bench(b, h, w, cin, cout, 3, 1, 1, tile=tiles.index(n), iters=5, precision=0))
    out.append(f"{name}: M={b*h*w} N={cout} K={9*cin}")
    for n in res:
        ms = min(res[n])
        out.append(f"  {n:16s} {ms:7.3f} ms  {flops/(ms*1e-3)/1e12:6.1f} TF   (" + " ".join(f"{m:.3f}" for m in res[n]) + ")")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/sbh_variants.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
