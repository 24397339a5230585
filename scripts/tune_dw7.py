"""Times the depthwise 7x7 kernels on the ConvNeXt-T shapes of the B=32 forward (pf_op_dwconv7x7_bench, random data):
one column per lane (variant 2), the column-blocked kernel (3) over nc x nb x strip height, the LDS-tile kernel (4), and the packed-fp32 forms (dw7_pk.hip):
streaming (5) over nc x nb x strip height, tile-in-parts (6) over channels per block x strip height.  TUNE_PK_ONLY=1: the default (4) and the packed forms only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
PK_ONLY = os.environ.get("TUNE_PK_ONLY", "0") == "1"
out = []


def t(variant, H, C, **kw):
    return min(ops.dwconv7x7_bench(variant, B, H, H, C, iters=20, **kw) for _ in range(3))


for (H, C) in ((80, 96), (40, 192), (20, 384), (10, 768)):
    mb = 8.0 * B * H * H * C / 1e6
    line = f"{H}x{H}x{C} ({mb:.1f} MB):"
    cur = t(4, H, C)
    line += f" default(4) {1e3 * cur:.1f} us {mb / cur / 1e3:.2f} TB/s |"
    if not PK_ONLY:
        base = t(2, H, C)
        res = [(t(3, H, C, nc=nc, nb=nb, th=th), nc, nb, th) for nc in (4, 2) for nb in (2, 3) for th in sorted({0, 5, 10, 20, 40, H} - {x for x in (5, 10, 20, 40) if x > H})]
        best = min(res)
        line += f" lane {1e3 * base:.1f} | best cb nc{best[1]} nb{best[2]} th{best[3]} {1e3 * best[0]:.1f} us |"
    res = [(t(5, H, C, nc=nc, nb=nb, th=th), nc, nb, th) for nc in (4, 2) for nb in (2, 3) for th in sorted({5, 10, 20, 40} - {x for x in (5, 10, 20, 40) if x > H})]
    best = min(res)
    line += f" best packed cb nc{best[1]} nb{best[2]} th{best[3]} {1e3 * best[0]:.1f} us {mb / best[0] / 1e3:.2f} TB/s [" + " ".join(f"nc{nc}nb{nb}th{th}:{1e3 * ms:.1f}" for ms, nc, nb, th in res) + "]"
    if H <= 40:
        res = [(t(6, H, C, nc=ch, th=th), ch, th) for ch in (32, 16) for th in (5, 10, 20) if th <= H and not (ch == 32 and (th == 20 or H > 20))]
        best = min(res)
        line += f" | best packed lds ch{best[1]} th{best[2]} {1e3 * best[0]:.1f} us {mb / best[0] / 1e3:.2f} TB/s [" + " ".join(f"ch{ch}th{th}:{1e3 * ms:.1f}" for ms, ch, th in res) + "]"
    out.append(line)
txt = "\n".join(out)
open(os.environ.get("TUNE_OUT", "gpurun_out/tune_dw7.txt"), "w").write(txt + "\n")
print(txt)
