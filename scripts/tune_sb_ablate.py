"""Linear split-GEMM tiles: where does a K step's time go, and two scheduling fixes -- TUNING BUILD only (PF_TUNING_BUILD=1 at build and run time).
Forms of four base tiles (igemm_sb_impl.h SB_ABL_PARAM):
  sbA16_* / sbA32_* / sbA48_*  no global loads of A / of B / of both (wrong results by construction, timing only);  sbA1_*  no split arithmetic while staging A
  sbI_*   (right results) the K step's (ky, kx, channel) position carried instead of two integer divisions per step (52 -> 29 SALU instructions per step)
  sbPI_*  (right results) that plus scheduling barriers: the next tile's loads in front of this tile's MFMAs, the split arithmetic behind them
Bit check of the right-result forms against their base tile, then ms / TF per (shape, tile).  Output: gpurun_out/sb_ablate.txt"""
import os, sys
import torch  # before the library: one HIP runtime per process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
tiles = ops.conv_tiles()
BASES = {"64x64": "sb64x64", "64x64f3": "sb64x64f3", "128x128": "sb128x128", "256x128w8": "sb256x128w8"}
FORMS = ["sbA16_", "sbA32_", "sbA48_", "sbA1_", "sbI_", "sbPI_"]
# (name, rows, K, N, bases)
SHAPES = [
    ("s3_qproj", B * 400, 320, 320, ["64x64", "64x64f3"]), ("s3_fc1", B * 400, 320, 1280, ["64x64", "64x64f3", "128x128"]), ("s3_fc2", B * 400, 1280, 320, ["64x64", "64x64f3"]),
    ("s4_fc1", B * 100, 512, 2048, ["64x64", "64x64f3"]), ("cnx2_pw2", B * 400, 1536, 384, ["64x64", "64x64f3"]),
    ("s2_fc1", B * 1600, 128, 512, ["128x128", "256x128w8"]), ("s1_fc1", B * 6400, 64, 256, ["128x128", "256x128w8"]), ("cnx1_pw1", B * 1600, 192, 768, ["128x128", "256x128w8"]),
]
out = []
torch.manual_seed(0)
for b, base in BASES.items():
    for f in ("sbI_", "sbPI_"):
        v = f + b
        if v not in tiles:
            out.append(f"bit check {v}: tile missing"); continue
        ok = True
        for rows, K, N in [(300, 64, 256), (257, 320, 128), (129, 1280, 320), (100, 96, 384)]:
            x = torch.randn(rows, K, device="cuda")
            w = torch.randn(N, K, device="cuda") * 0.05
            bias = torch.randn(N, device="cuda")
            x4, w4 = x.reshape(1, rows, 1, K), w.reshape(N, K, 1, 1)  # splitk=False: the tuning forms never split K, the base tile would on the deep-K case
            y0 = ops.conv2d(x4, w4, bias, tile=tiles.index(base), splitk=False)
            y1 = ops.conv2d(x4, w4, bias, tile=tiles.index(v), splitk=False)
            ok = ok and bool(torch.equal(y0, y1))
        out.append(f"bit check {v} vs {base}: {'identical' if ok else 'DIFFERS'}")
for name, rows, K, N, bases in SHAPES:
    flops = 2.0 * rows * K * N
    out.append(f"{name}: M={rows} N={N} K={K}")
    for b in bases:
        names = [BASES[b]] + [f + b for f in FORMS]
        res = {n: [] for n in names if n in tiles}
        for rep in range(2):
            for n in res:
                res[n].append(ops.conv2d_bench(1, rows, 1, K, N, 1, 1, 0, tile=tiles.index(n), iters=20, precision=0))
        base_ms = min(res[BASES[b]])
        for n in res:
            ms = min(res[n])
            out.append(f"  {n:18s} {ms*1e3:8.1f} us  {flops/(ms*1e-3)/1e12:6.1f} TF  {ms/base_ms*100:6.1f} %")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/sb_ablate.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
