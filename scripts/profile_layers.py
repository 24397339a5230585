"""Per-layer table from the engine's HIP-event profiler: one profiled forward, grouped by GEMM shape."""
import argparse, collections, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import PerspectiveFields
from perspectivefields_amd.synth import synthetic_image

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--version", default="Paramnet-360Cities-edina-centered")
ap.add_argument("--out", default="gpurun_out/layers.txt"); ap.add_argument("--precision", default="fp32"); ap.add_argument("--autotune", type=int, default=0)
a = ap.parse_args()
m = PerspectiveFields(a.version, weights="synthetic:0", precision=a.precision).eval().cuda()
eng = m._get_engine()
if a.autotune: eng.autotune(a.batch)
x = torch.from_numpy(np.stack([m.aug.apply_image(synthetic_image(640, 640, i % 4)) for i in range(a.batch)])).cuda()
for _ in range(2): eng.forward(x)
torch.cuda.synchronize()
eng.profile_begin(); eng.forward(x); torch.cuda.synchronize(); tot = eng.profile_end(); recs = eng.profile_records()
lines = []
allms = sum(v["ms"] for v in tot.values())
lines.append(f"batch {a.batch} precision {a.precision} autotune {a.autotune}: profiled classes total {allms:.2f} ms")
for k, v in tot.items():
    if v["launches"]:
        unit = "TFLOP/s" if k in ("igemm", "igemm_sb", "attention") else "GB/s"
        rate = v["work"] / (v["ms"] * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
        lines.append(f"  {k:16s} {v['launches']:4d} launches {v['ms']:8.3f} ms  {100*v['ms']/allms:5.1f}%  {rate:8.1f} {unit}")
g = collections.OrderedDict()
for cls, work, ms, mnk in recs:
    if cls not in ("igemm", "igemm_sb"): continue
    e = g.setdefault(mnk + (cls,), [0, 0.0, 0.0]); e[0] += 1; e[1] += ms; e[2] += work
lines.append("igemm by shape (M, N, K, KH): launches, total ms, TFLOP/s")
for mnk, (n, ms, work) in sorted(g.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"  M={mnk[0]:8d} N={mnk[1]:5d} K={mnk[2]:6d} KH={mnk[3]} {mnk[4]:8s} x{n:3d}  {ms:8.3f} ms  {work/(ms*1e-3)/1e12:7.1f} TF")
# HBM-bound classes grouped by launch size (algorithmic bytes per launch)
h = collections.OrderedDict()
for cls, work, ms, mnk in recs:
    if cls in ("igemm", "igemm_sb"): continue
    e = h.setdefault((cls, int(work)), [0, 0.0]); e[0] += 1; e[1] += ms
lines.append("other classes by launch size: class, MB (attention: MFLOP) per launch, launches, avg us, GB/s (attention: GFLOP/s)")
for (cls, work), (n, ms) in sorted(h.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    lines.append(f"  {cls:16s} {work/1e6:9.1f} MB x{n:3d}  {1e3*ms/n:8.1f} us  {work*n/(ms*1e-3)/1e9:8.1f} GB/s")
open(a.out, "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
