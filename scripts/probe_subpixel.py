"""VERDICT r05 item 6, measured: what the sub-pixel (phase) forms of conv_fuse_conv1 / conv_fuse_conv0 would cost at best.

By linearity, conv3x3(bilinear_x2(x)) on the 2H x 2W map = four phase-specific 3x3 convs on the H x W map (4 x Cout output channels, same FLOPs), which makes the layer
Winograd-eligible -- plus border corrections (the high-resolution zero padding is not the low-resolution one: a 2-pixel ring of outputs) and, for conv0, the 64 -> 64
low-level branch at full resolution.  This script times ONLY the main terms with the shipped kernels (random data, B = 64 = 32 images x 2 heads in one launch, min of 3 x 10):
a lower bound of the rewrite, to be compared with the fused direct kernels' times in the forward (gravity_head.py:170-176)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

names = ops.conv_tiles()
wd = names.index("wino256x64d")
t = lambda B, H, Ci, Co, tile: min(ops.conv2d_bench(B, H, H, Ci, Co, 3, 1, 1, tile=tile, iters=10) for _ in range(3))
out = []
ms = t(64, 160, 64, 128, wd)
out.append(f"conv1 as 4 phase convs: Winograd 64 -> 128 on 160^2 (both heads): {ms:.3f} ms   (fused direct kernel in the forward: 1.18 ms; target <= 0.85 ms incl. border corrections)")
best = min((t(64, 160, 64, 128, names.index(n)), n) for n in names if n.startswith("sbh") and not n.startswith(("sbhA", "sbhLA", "sbhDMA", "sbhREG", "sbhV")))
out.append(f"   same shape on the best direct halo tile ({best[1]}): {best[0]:.3f} ms")
a = t(64, 80, 256, 256, wd)
bb = min((t(64, 160, 64, 64, names.index(n)), n) for n in names if n.startswith("sbh") and not n.startswith(("sbhA", "sbhLA", "sbhDMA", "sbhREG", "sbhV")))
out.append(f"conv0 as 4 phase convs: Winograd 256 -> 256 on 80^2 (both heads): {a:.3f} ms + low-level branch 64 -> 64 on 160^2, best direct tile ({bb[1]}): {bb[0]:.3f} ms = {a + bb[0]:.3f} ms "
           f"+ one more pass over the 160^2 x 64 map (0.42 GB: ~0.1 ms) + border corrections   (fused direct kernel in the forward: 1.66 ms)")
txt = "\n".join(out)
open("gpurun_out/r06_subpixel_probe.txt", "w").write(txt + "\n")
print(txt)
