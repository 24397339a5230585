"""Ablation timing of the 3x3 halo kernel's dominant tile (sbh256x64w8) -- TUNING BUILD only (PF_TUNING_BUILD=1 at build and run time).
Tiles "sbhA<mask>" run the same kernel with one cost removed (igemm_sbh.hip SBH_ABL_PARAM; results are wrong by construction):
  1 = no split arithmetic in the halo staging, 2 = no wh 2^-11 scaling of the weight fragment, 4 = no barriers in the K loop (spills 13 registers),
  8 = half the LDS fragment reads, 16 / 32 (sbhAa / sbhAb; 48 = both) = no global loads of the halo / of the weights in the K loop; sums combine.
  sbhLA0 / sbhLA2 (right results): next halo chunk's loads issued in tap step 0 / 2 instead of 4.
Output: gpurun_out/sbh_ablate.txt -- ms and fp32-equivalent TFLOP/s per form, three repeats interleaved."""
import os, sys, time
import torch  # before the library: libpf_hip.so must bind to the HIP runtime PyTorch loads (one runtime per process)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
SHAPES = [("rcu80", B, 80, 80, 256, 256), ("rcu40", B, 40, 40, 256, 256), ("fold_c1", B, 80, 80, 64, 256)]
tiles = ops.conv_tiles()
want = os.environ.get("ABL_TILES", "sbh256x64w8,sbhA1,sbhA2,sbhA3,sbhA4,sbhA8,sbhA48,sbhAa,sbhAb,sbhA12,sbhA11,sbhA15,sbhA63,sbhLA0,sbhLA2,sbh128x64").split(",")
ids = [(n, tiles.index(n)) for n in want if n in tiles]
if not any(n.startswith("sbhA") for n, _ in ids):
    raise SystemExit("no ablation tiles in this library: build and run with PF_TUNING_BUILD=1")
out = []
# the "real" variants (not sbhA*) must give the bits of the base tile
torch.manual_seed(0)
base_id = tiles.index("sbh256x64w8")
for n, t in ids:
    if n.startswith("sbhA") or not n.startswith("sbh") or n == "sbh256x64w8" or n in ("sbh128x64", "sbh128x128"):
        continue
    ok = True
    for (b, h, w, cin, cout) in [(2, 40, 40, 64, 128), (1, 33, 47, 32, 96), (1, 16, 16, 96, 64)]:
        x = torch.randn(b, h, w, cin, device="cuda")
        wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        bias = torch.randn(cout, device="cuda")
        y0 = ops.conv2d(x, wt, bias, stride=1, pad=1, tile=base_id)
        y1 = ops.conv2d(x, wt, bias, stride=1, pad=1, tile=t)
        ok = ok and bool(torch.equal(y0, y1))
    out.append(f"bit check {n}: {'identical to sbh256x64w8' if ok else 'DIFFERS'}")
for name, b, h, w, cin, cout in SHAPES:
    flops = 2.0 * b * h * w * cout * 9 * cin
    res = {n: [] for n, _ in ids}
    for rep in range(3):
        for n, t in ids:
            ms = ops.conv2d_bench(b, h, w, cin, cout, 3, 1, 1, tile=t, iters=10, precision=0)
            res[n].append(ms)
    base = min(res["sbh256x64w8"])
    out.append(f"{name}: M={b*h*w} N={cout} K={9*cin}")
    for n, _ in ids:
        ms = min(res[n])
        out.append(f"  {n:12s} {ms:7.3f} ms  {flops/(ms*1e-3)/1e12:6.1f} TF  {ms/base*100:6.1f} % of sbh256x64w8   (repeats: " + " ".join(f"{m:.3f}" for m in res[n]) + ")")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/sbh_ablate.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
