"""which yardstick do the depthwise 7x7 launches of ConvNeXt-T (convnext.py:30-33,46-52) belong to?

The rocprofv3 FETCH_SIZE / WRITE_SIZE counters sit on the L2's fabric side and count Infinity-Cache hits like HBM reads (MI355X_MICROARCH.md, "HBM"), so they cannot
tell whether a 20 - 160 MB map written by the previous kernel is served from the 256 MiB Infinity Cache.  Timing can: the same launch WARM (one buffer pair re-used by
every launch: the state inside the forward, where the previous kernel has just written the input) and COLD (cycling through enough buffer pairs that a launch's input
was last touched > 512 MB of traffic ago), next to a plain float4 copy of the same bytes (pf_op_dwconv7x7_bench variant 100) -- what this memory system gives ANY
launch of that size, fixed cost (launch, first byte, drain) included.  B = 32, min of 3 x 30 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
rows = []
tot = {"dw_warm": 0.0, "dw_cold": 0.0, "cp_warm": 0.0, "cp_cold": 0.0, "mb": 0.0}
launches = {80: 3, 40: 3, 20: 9, 10: 3}


def t(variant, H, C, cold):
    mb2 = 8.0 * B * H * H * C / 1e6
    if cold:
        os.environ["PF_DW7_BENCH_COLD"] = str(max(2, int(640.0 / mb2) + 1))
    else:
        os.environ.pop("PF_DW7_BENCH_COLD", None)
    return min(ops.dwconv7x7_bench(variant, B, H, H, C, iters=30) for _ in range(3))


for (H, C) in ((80, 96), (40, 192), (20, 384), (10, 768)):
    mb = 8.0 * B * H * H * C / 1e6
    dw_w, dw_c, cp_w, cp_c = t(7, H, C, False), t(7, H, C, True), t(100, H, C, False), t(100, H, C, True)
    rows.append((H, C, mb, dw_w, dw_c, cp_w, cp_c))
    n = launches[H]
    tot["mb"] += n * mb
    for k, v in (("dw_warm", dw_w), ("dw_cold", dw_c), ("cp_warm", cp_w), ("cp_cold", cp_c)):
        tot[k] += n * v
lines = ["| map (MB in + out) | dw7x7 warm us (TB/s) | dw7x7 cold us (TB/s) | copy warm us (TB/s) | copy cold us (TB/s) | dw7 / copy, warm | dw7 / copy, cold |", "|---|---|---|---|---|---|---|"]
f = lambda mb, ms: f"{1e3 * ms:.1f} ({mb / ms / 1e3:.2f})"
for H, C, mb, dw_w, dw_c, cp_w, cp_c in rows:
    lines.append(f"| {H}^2 x {C} ({mb:.1f}) | {f(mb, dw_w)} | {f(mb, dw_c)} | {f(mb, cp_w)} | {f(mb, cp_c)} | {cp_w / dw_w:.2f} | {cp_c / dw_c:.2f} |")
lines.append(f"| the 18 launches of a forward ({tot['mb']:.0f}) | {f(tot['mb'], tot['dw_warm'])} | {f(tot['mb'], tot['dw_cold'])} | {f(tot['mb'], tot['cp_warm'])} | {f(tot['mb'], tot['cp_cold'])} | "
             f"{tot['cp_warm'] / tot['dw_warm']:.2f} | {tot['cp_cold'] / tot['dw_cold']:.2f} |")
txt = "\n".join(lines)
open(os.environ.get("TUNE_OUT", "gpurun_out/r06_dw7_yardstick.md"), "w").write(txt + "\n")
print(txt)
