"""Row-resident GEMM ("sbr" tiles, csrc/rr_gemm.hip) against the LDS-tiled linear tiles on the linear / kernel == stride conv shapes of the B = 32 forward.
ms and TF per (shape, tile); the best LDS tile vs the best sbr tile per shape.  Output: gpurun_out/tune_rr.txt"""
import os, sys
import torch  # before the library: one HIP runtime per process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
tiles = ops.conv_tiles()
LDS = ["sb64x64", "sb64x64f2", "sb64x64f3", "sb128x64", "sb128x64f2", "sb128x128", "sb256x128w8"]
RR = ["sbr128x160", "sbr128x128"]
# (name, B, H, W, Cin, Cout, k)   linear layers: H = rows per image, W = 1
SHAPES = [
    ("mit3 q/proj", B, 400, 1, 320, 320, 1), ("mit3 fc1", B, 400, 1, 320, 1280, 1), ("mit3 fc2", B, 400, 1, 1280, 320, 1), ("mit3 kv", B, 100, 1, 320, 640, 1), ("mit3 sr 2x2s2", B, 20, 20, 320, 320, 2),
    ("cnx3 pw1", B, 400, 1, 384, 1536, 1), ("cnx3 pw2", B, 400, 1, 1536, 384, 1),
    ("mit2 q/proj", B, 1600, 1, 128, 128, 1), ("mit2 fc1", B, 1600, 1, 128, 512, 1), ("mit2 fc2", B, 1600, 1, 512, 128, 1), ("mit2 kv", B, 100, 1, 128, 256, 1), ("mit2 sr 4x4s4", B, 40, 40, 128, 128, 4),
    ("mit4 q/proj", B, 100, 1, 512, 512, 1), ("mit4 kv", B, 100, 1, 512, 1024, 1), ("mit4 fc1", B, 100, 1, 512, 2048, 1), ("mit4 fc2", B, 100, 1, 2048, 512, 1),
    ("cnx2 pw1", B, 1600, 1, 192, 768, 1), ("cnx2 pw2", B, 1600, 1, 768, 192, 1), ("cnx4 pw1", B, 100, 1, 768, 3072, 1), ("cnx4 pw2", B, 100, 1, 3072, 768, 1),
    ("mit1 q/proj", B, 6400, 1, 64, 64, 1), ("mit1 sr 8x8s8", B, 80, 80, 64, 64, 8), ("cnx ds 2x2s2 96", B, 80, 80, 96, 192, 2), ("cnx ds 2x2s2 192", B, 40, 40, 192, 384, 2), ("cnx ds 2x2s2 384", B, 20, 20, 384, 768, 2),
]
out = []
tot_l = tot_r = 0.0
for name, b, h, w, cin, cout, k in SHAPES:
    M = b * (h // k) * (w // k if w > 1 else 1)
    flops = 2.0 * M * cout * cin * k * k
    res = {}
    for rep in range(2):
        for n in LDS + RR:
            if n not in tiles:
                continue
            ms = ops.conv2d_bench(b, h, w, cin, cout, k, k, 0, tile=tiles.index(n), iters=20, precision=0)
            if ms > 0:
                res[n] = min(res.get(n, 1e9), ms)
    bl = min((res[n], n) for n in LDS if n in res)
    br = min(((res[n], n) for n in RR if n in res), default=(float("nan"), "-"))
    tot_l += bl[0]; tot_r += min(bl[0], br[0]) if br[1] != "-" else bl[0]
    out.append(f"{name:18s} M={M:6d} K={cin * k * k:5d} N={cout:5d}   LDS tiles: " + " ".join(f"{n[2:]} {res[n] * 1e3:6.1f}" for n in LDS if n in res))
    out.append(f"{'':18s} best LDS {bl[1]:12s} {bl[0] * 1e3:7.1f} us {flops / (bl[0] * 1e-3) / 1e12:6.1f} TF | " + " ".join(f"{n} {res[n] * 1e3:6.1f} us" for n in RR if n in res) +
               (f" | best sbr {br[0] * 1e3:7.1f} us {flops / (br[0] * 1e-3) / 1e12:6.1f} TF  = {br[0] / bl[0] * 100:5.1f} %" if br[1] != "-" else " | sbr: not eligible"))
out.append(f"sum over the listed shapes (one launch each): best LDS tile {tot_l * 1e3:.1f} us, best of both {tot_r * 1e3:.1f} us")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/tune_rr.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
