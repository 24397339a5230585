"""Winograd F(2x2, 3x3) tiles (wino.hip) against the direct halo tiles on the 3x3 / stride-1 shapes of the decoders (pf_op_conv2d_bench, one head, random
data, best of 3 interleaved repeats).  Output: gpurun_out/tune_wino.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
SHAPES = [("rcu80", B, 80, 80, 256, 256), ("rcu40", B, 40, 40, 256, 256), ("rcu20", B, 20, 20, 256, 256), ("rcu10", B, 10, 10, 256, 256),
          ("fold_c1", B, 80, 80, 64, 256), ("fold_c2", B, 40, 40, 128, 256), ("rcu80_2heads", 2 * B, 80, 80, 256, 256)]
names = ops.conv_tiles()
cand = [n for n in ("wino256x64d", "wino256x64c", "sbh256x64w8", "sbh128x128") if n in names]
out = [f"B={B}; ms per launch (best of 3 x 5 launches) and algorithmic TFLOP/s (2 M N 9 Cin)"]
for name, b, h, w, cin, cout in SHAPES:
    flops = 2.0 * b * h * w * cout * 9 * cin
    best = {n: 1e9 for n in cand}
    for rep in range(3):
        for n in cand:
            ms = ops.conv2d_bench(b, h, w, cin, cout, 3, 1, 1, tile=names.index(n), iters=5)
            if ms > 0:
                best[n] = min(best[n], ms)
    out.append(f"{name:13s} M={b*h*w:7d} Cin={cin:3d} Cout={cout:3d} | " + " | ".join(f"{n} {best[n]:7.3f} ms {flops / (best[n] * 1e-3) / 1e12:6.1f} TF" for n in cand if best[n] < 1e9))
txt = "\n".join(out)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/tune_wino.txt", "w").write(txt + "\n")
print(txt)
