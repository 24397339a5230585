import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops
B = 32
out = []
for (H, C) in [(80, 256), (40, 512), (20, 1280), (10, 2048)]:
    mb = 2.0 * B * H * H * C * 4 / 1e6
    row = f"dw3x3 {H}x{H}x{C} ({mb:.0f} MB):"
    for v in (0, 1, 2, 3, 4, 51, 52, 99):
        ms = ops.dwconv3x3_bench(v, B, H, H, C, iters=20)
        row += f"  v{v}: {ms*1000:7.1f} us {mb/ms/1e3:5.2f} TB/s"
    out.append(row)
print("\n".join(out)); open("gpurun_out/tune_dw.txt", "w").write("\n".join(out) + "\n")
