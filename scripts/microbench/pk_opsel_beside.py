"""Driver of pk_opsel_beside.hip: every packed-fp32 operand form alone and beside this library's B = 32 forward (background thread on another stream); a launch "differs"
when the packed chain's result is not bit-identical to the scalar chain of the same launch.  Output: gpurun_out/pk_opsel_beside.txt"""
import ctypes, os, sys, threading
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from perspectivefields_amd import PerspectiveFields
from perspectivefields_amd.synth import synthetic_image

lib = ctypes.CDLL(os.path.join(HERE, "libpk_opsel_beside.so"))
lib.pk_form_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
FORMS = ["0: pk_fma  w = src1, both lanes x w.hi   op_sel:[0,1,0] op_sel_hi:[1,1,1]", "1: pk_fma  w = src0, both lanes x w.hi   op_sel:[1,0,0] op_sel_hi:[1,1,1]",
         "2: pk_fma  w = src1, both lanes x w.lo   op_sel_hi:[1,0,1]", "3: pk_fma  w = src1, halves swapped      op_sel:[0,1,0] op_sel_hi:[1,0,1]",
         "4: pk_mul  w = src1, both lanes x w.hi   op_sel:[0,1] op_sel_hi:[1,1]", "5: pk_add  src1 halves swapped          op_sel:[0,1] op_sel_hi:[1,0]",
         "6: pk_fma  src2 halves swapped          op_sel:[0,0,1] op_sel_hi:[1,1,0]",
         "7: pk_add  d, x, x (same pair twice)    op_sel:[0,1] op_sel_hi:[1,0]   <- hipcc's horizontal-sum form"]
LAUNCHES = {7: 200}   # the form the product used to contain: 200 launches, 40 for the others
m = PerspectiveFields("Paramnet-360Cities-edina-centered", weights="synthetic:0").eval().cuda()
eng = m._get_engine()
xb = torch.from_numpy(np.stack([m.aug.apply_image(synthetic_image(80, 100, seed=i)) for i in range(32)])).cuda()
eng.forward(xb); torch.cuda.synchronize()
g = torch.Generator().manual_seed(1)
N = 1 << 20
a = torch.randn(N + 4096, 2, generator=g).cuda(); w = (torch.randn(N + 4096, 2, generator=g) * 0.15).cuda()
BLOCKS, ITERS = 2048, 24
out_pk = torch.empty(BLOCKS * 256, 2, device="cuda"); out_ref = torch.empty_like(out_pk)
stop = threading.Event(); bg = torch.cuda.Stream(); side = torch.cuda.Stream()

def background():
    with torch.cuda.stream(bg):
        while not stop.is_set():
            eng.forward(xb); bg.synchronize()

lines = []
for phase in ("alone", "beside this library's B=32 forward"):
    if phase != "alone":
        t = threading.Thread(target=background); t.start()
    for f, name in enumerate(FORMS):
        bad, pat = 0, ""
        with torch.cuda.stream(side):
            for it in range(LAUNCHES.get(f, 40)):
                rc = lib.pk_form_launch(f, a.data_ptr(), w.data_ptr(), out_pk.data_ptr(), out_ref.data_ptr(), BLOCKS, ITERS, side.cuda_stream)
                assert rc == 0
                side.synchronize()
                if not torch.equal(out_pk, out_ref):
                    bad += 1
                    if not pat:
                        d = (out_pk != out_ref)
                        idx = d.nonzero()
                        lanes = sorted(set((idx[:, 0] % 64).tolist()))
                        pat = f" first: {idx.shape[0]} of {out_pk.numel()} values; element (0 = lo, 1 = hi) {sorted(set(idx[:, 1].tolist()))}; lanes {lanes[0]}..{lanes[-1]} ({len(lanes)} distinct); max|d| {float((out_pk - out_ref).abs().max()):.2e}"
        lines.append(f"[{phase}] form {name}: {bad}/{LAUNCHES.get(f, 40)} launches differ{pat}")
        print(lines[-1], flush=True)
    if phase != "alone":
        stop.set(); t.join()
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/pk_opsel_beside.txt", "w").write("\n".join(lines) + "\n")
