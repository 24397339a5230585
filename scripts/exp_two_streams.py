"""Experiment: the B = 32 step as TWO half-batches on two streams (two engine handles, own workspaces) against one B = 32 forward.  Hypothesis: the small-M launches of
the backbone / ParamNet (~1000 blocks, latency-bound) of the two halves overlap with each other, and -- once the halves drift apart -- with the other half's decoder.
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

def two(n, skew=False):
    h = B // 2
    for it in range(n):
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                pg, pl, par = engs[k].forward(x[k * h:(k + 1) * h])
                engs[k].postprocess_batch(pg, pl, sizes[:h])

def timed(fn, n=10):
    fn(3); torch.cuda.synchronize()
    t = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

for rep in range(2):
    t1 = timed(one)
    t2 = timed(two)
    out.append(f"rep {rep}: one B={B} forward {t1 * 1e3:.2f} ms ({B / t1:.0f} img/s) | two B={B // 2} halves on two streams {t2 * 1e3:.2f} ms ({B / t2:.0f} img/s)  -> {t1 / t2:.3f}x")
# results identical?
pg, pl, par = engs[0].forward(x)
with torch.cuda.stream(streams[1]):
    pg2, pl2, par2 = engs[1].forward(x[B // 2:])
torch.cuda.synchronize()
out.append(f"second half vs full-batch forward: max |d pred_gravity| {float((pg[B // 2:] - pg2).abs().max()):.2e}, params {float((par[B // 2:] - par2).abs().max()):.2e}")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/exp_two_streams.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
