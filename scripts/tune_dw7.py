"""Times the depthwise 7x7 kernels on the ConvNeXt-T shapes of the B=32 forward (pf_op_dwconv7x7_bench, random data):
one column per lane (variant 2) and the column-blocked kernel (variant 3) over nc x nb x strip height."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
out = []
for (H, C) in ((80, 96), (40, 192), (20, 384), (10, 768)):
    mb = 8.0 * B * H * H * C / 1e6
    base = ops.dwconv7x7_bench(2, B, H, H, C, iters=20)
    res = []
    for nc in (4, 2):
        for nb in (2, 3):
            for th in sorted({0, 5, 10, 20, 40, H} - {t for t in (5, 10, 20, 40) if t > H}):
                ms = ops.dwconv7x7_bench(3, B, H, H, C, nc=nc, nb=nb, th=th, iters=20)
                res.append((ms, nc, nb, th))
    best = min(res)
    out.append(f"{H}x{H}x{C} ({mb:.1f} MB): lane {1e3*base:.1f} us {mb/base/1e3:.0f} GB/s | best cb nc{best[1]} nb{best[2]} th{best[3]} {1e3*best[0]:.1f} us {mb/best[0]/1e3:.0f} GB/s | " +
               " ".join(f"nc{nc}nb{nb}th{th}:{1e3*ms:.1f}" for ms, nc, nb, th in res))
txt = "\n".join(out)
open(os.environ.get("TUNE_OUT", "gpurun_out/tune_dw7.txt"), "w").write(txt + "\n")
print(txt)
