"""Experiment: step-level pipelining on two streams.  (a) r03: the B = 32 step as TWO half-batches on two streams (rejected: half batches run their launches less
efficiently).  (b) r04: FULL batches alternating between two engines on two streams (two workspaces, weights twice), with and without the deferred ParamNet branch --
do the launch gaps of one forward get filled by the other's kernels beyond what the deferred branch already harvests?
Output: gpurun_out/exp_two_streams.txt"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import PerspectiveFields
from perspectivefields_amd.synth import synthetic_image

V = "Paramnet-360Cities-edina-centered"
B = int(os.environ.get("EXP_B", "32"))
models = [PerspectiveFields(V, weights="synthetic:0").eval().cuda() for _ in range(2)]
engs = [m._get_engine() for m in models]
x = torch.from_numpy(np.stack([models[0].aug.apply_image(synthetic_image(640, 640, i % 4)) for i in range(B)])).cuda()
sizes = [(640, 640)] * B
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
out = []

def one(n):
    for _ in range(n):
        pg, pl, par = engs[0].forward(x)
        engs[0].postprocess_batch(pg, pl, sizes)
    engs[0].join_params()

def two_full(n):
    for it in range(n):
        k = it & 1
        with torch.cuda.stream(streams[k]):
            pg, pl, par = engs[k].forward(x)
            engs[k].postprocess_batch(pg, pl, sizes)
    for k in (0, 1):
        with torch.cuda.stream(streams[k]):
            engs[k].join_params()

def timed(fn, n=10):
    fn(4); torch.cuda.synchronize()
    t = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

for defer in (False, True):
    for e in engs:
        e.set_defer_params(defer)
    for rep in range(2):
        t1 = timed(one)
        t2 = timed(two_full)
        out.append(f"defer {int(defer)} rep {rep}: one stream {t1 * 1e3:.2f} ms/step ({B / t1:.0f} img/s) | full batches alternating on two streams {t2 * 1e3:.2f} ms/step ({B / t2:.0f} img/s)  -> {t1 / t2:.3f}x")
for e in engs:
    e.set_defer_params(False)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/exp_two_streams.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
