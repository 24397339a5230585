"""Repro / diagnosis: the deferred ParamNet branch against joined forwards, many trials, per-forward mismatch report (index, max |d| of the parameters, fields equal?).
Env switches (PF_DW7_VARIANT, ...) select the kernels under test."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import PerspectiveFields
from perspectivefields_amd.synth import synthetic_image

V = os.environ.get("REPRO_VERSION", "Paramnet-360Cities-edina-centered")
m = PerspectiveFields(V, weights="synthetic:0").eval().cuda()
eng = m._get_engine()
sizes = tuple(int(s) for s in os.environ.get("REPRO_B", "16,16,5,16").split(","))
xs = [torch.from_numpy(np.stack([m.aug.apply_image(synthetic_image(80, 100, seed=900 + 20 * j + i)) for i in range(n)])).cuda() for j, n in enumerate(sizes)]
ref = []
for x in xs:
    ref.append([t.clone() for t in eng.forward(x)])
    torch.cuda.synchronize()
# joined forwards are deterministic?
for rep in range(3):
    for i, x in enumerate(xs):
        o = eng.forward(x); torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip(o, ref[i])):
            print(f"JOINED rep {rep} forward {i}: differs from the first joined run, params max|d| {float((o[2] - ref[i][2]).abs().max()):.3e}")
bad = 0
eng.set_defer_params(True)
for trial in range(int(os.environ.get("REPRO_TRIALS", "8"))):
    outs, snaps = [], []
    for i, x in enumerate(xs):
        outs.append(eng.forward(x))
        if i > 0:
            snaps.append(outs[i - 1][2].clone())
    eng.join_params()
    snaps.append(outs[-1][2].clone())
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, ref)):
        f = torch.equal(o[0], r[0]) and torch.equal(o[1], r[1])
        p = torch.equal(snaps[i], r[2]); q = torch.equal(o[2], r[2])
        if not (f and p and q):
            bad += 1
            print(f"DEFERRED trial {trial} forward {i} (B={sizes[i]}): fields equal {f}, snapshot equal {p} (max|d| {float((snaps[i] - r[2]).abs().max()):.3e}), final tensor equal {q} (max|d| {float((o[2] - r[2]).abs().max()):.3e}), rows differing {int((snaps[i] != r[2]).any(1).sum())}")
eng.set_defer_params(False)
print(f"variant {os.environ.get('PF_DW7_VARIANT', 'default')} ch {os.environ.get('PF_DW7_PK_CH', '-')} sizes {sizes}: {bad} mismatching forwards")
