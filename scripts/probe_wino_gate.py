"""Which 3x3 / stride-1 shapes of the decoders should take the Winograd kernel?  Winograd vs the best direct halo tile per shape (random data, both heads in one launch, min of 3 x 10)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

names = ops.conv_tiles()
wd = names.index("wino256x64d")
direct = [n for n in names if n.startswith("sbh") and not n.startswith(("sbhA", "sbhLA", "sbhDMA", "sbhREG", "sbhV"))]
t = lambda B, H, Ci, Co, tile: min(ops.conv2d_bench(B, H, H, Ci, Co, 3, 1, 1, tile=tile, iters=10) for _ in range(3))
lines = []
for (B, H, Ci, Co, what) in ((64, 80, 64, 256, "folded first conv @80^2"), (64, 40, 128, 256, "folded first conv @40^2"), (64, 80, 256, 256, "RCU @80^2"), (64, 40, 256, 256, "RCU @40^2"),
                             (64, 20, 256, 256, "RCU @20^2"), (16, 80, 256, 256, "RCU @80^2, B = 8"), (16, 40, 256, 256, "RCU @40^2, B = 8"), (2, 80, 256, 256, "RCU @80^2, B = 1"), (2, 40, 256, 256, "RCU @40^2, B = 1")):
    w = t(B, H, Ci, Co, wd)
    d = min((t(B, H, Ci, Co, names.index(n)), n) for n in direct)
    lines.append(f"{what:28s} B{B:3d} {H}^2 {Ci}->{Co}: winograd {1e3 * w:7.1f} us | best direct {1e3 * d[0]:7.1f} us ({d[1]})  -> {'winograd' if w < d[0] else 'DIRECT'} {max(w, d[0]) / min(w, d[0]):.2f}x")
txt = "\n".join(lines)
open("gpurun_out/r06_wino_gate.txt", "w").write(txt + "\n")
print(txt)
