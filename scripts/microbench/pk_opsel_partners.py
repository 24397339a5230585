"""Which kernels of this library make the src1-high-half packed-fp32 forms fail?  pk_opsel_beside.hip's forms 3 / 4 / 5 (the ones that reproduce in the chain
microbenchmark) beside ONE kernel class at a time, looped by a background thread through the library's bench entry points.  Output: gpurun_out/pk_opsel_partners.txt"""
import ctypes, os, sys, threading
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from perspectivefields_amd import ops

lib = ctypes.CDLL(os.path.join(HERE, "libpk_opsel_beside.so"))
lib.pk_form_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
g = torch.Generator().manual_seed(1)
N = 1 << 20
a = torch.randn(N + 4096, 2, generator=g).cuda(); w = (torch.randn(N + 4096, 2, generator=g) * 0.15).cuda()
BLOCKS, ITERS = 2048, 24
out_pk = torch.empty(BLOCKS * 256, 2, device="cuda"); out_ref = torch.empty_like(out_pk)
side = torch.cuda.Stream()
A16 = torch.randn(8192, 8192, device="cuda", dtype=torch.float16); B16 = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
q = torch.randn(32, 400, 320, device="cuda"); kv = torch.randn(32, 100, 640, device="cuda")
mm_stream = torch.cuda.Stream()

def mm_loop():
    with torch.cuda.stream(mm_stream):
        for _ in range(20):
            torch.mm(A16, B16)
        mm_stream.synchronize()

def attn_loop():
    with torch.cuda.stream(mm_stream):
        ops.sr_attention_variant(q, kv, 5, 1, iters=300)

t = ops.conv_tiles()


def forward_partner():
    """PK_FORWARD=<model version>: the partner is this library's forward (engine switches come from the environment of the process)"""
    import numpy as np
    from perspectivefields_amd import PerspectiveFields
    from perspectivefields_amd.synth import synthetic_image
    m = PerspectiveFields(os.environ["PK_FORWARD"], weights="synthetic:0").eval().cuda()
    eng = m._get_engine()
    xb = torch.from_numpy(np.stack([m.aug.apply_image(synthetic_image(80, 100, seed=i)) for i in range(32)])).cuda()
    eng.forward(xb); torch.cuda.synchronize()
    def run():
        with torch.cuda.stream(mm_stream):
            eng.forward(xb); mm_stream.synchronize()
    return run


PARTNERS = [
    ("nothing", None),
    ("3x3 halo conv 256->256 @80^2 B=8 (igemm_sbh, 8 waves, LDS-DMA weight ring, MFMA f16)", lambda: ops.conv2d_bench(8, 80, 80, 256, 256, 3, 1, 1, iters=40)),
    ("1x1 linear M=12800 K=320 N=1280 (igemm_sb 128x128, MFMA f16)", lambda: ops.conv2d_bench(32, 20, 20, 320, 1280, 1, 1, 0, iters=300)),
    ("1x1 linear, exact bf16 split (precision 3)", lambda: ops.conv2d_bench(32, 20, 20, 320, 1280, 1, 1, 0, iters=200, precision=3)),
    ("split-f16 attention (MFMA f16, no LDS-DMA)", attn_loop),
    ("depthwise 3x3 + GELU (VALU, compiler-packed fp32)", lambda: ops.dwconv3x3_bench(-1, 32, 40, 40, 512, iters=300)),
    ("scalar depthwise 7x7 (VALU only)", lambda: ops.dwconv7x7_bench(3, 32, 40, 40, 192, iters=300)),
    ("rocBLAS fp16 GEMM 8192^3", mm_loop),
]
if os.environ.get("PK_OPS"):
    # the same kernel classes through the op entry points ON A TORCH STREAM (the bench entry points above launch on the legacy null stream, which torch's streams may
    # simply serialise with): each call re-uploads its weights, so the GPU duty cycle is low -- big shapes
    import math
    gg = torch.Generator().manual_seed(3)
    x80 = torch.randn(32, 80, 80, 256, generator=gg).cuda(); w33 = torch.randn(256, 256, 3, 3, generator=gg) / 48.0
    xl = torch.randn(204800, 128, generator=gg).cuda(); wl = torch.randn(512, 128, generator=gg) / 11.3; bl = torch.zeros(512); gl = torch.ones(128); bel = torch.zeros(128)
    xln = torch.randn(204800, 128, generator=gg).cuda()
    xd = torch.randn(32, 80, 80, 512, generator=gg).cuda(); wd = torch.randn(512, 1, 3, 3, generator=gg) * 0.3; bd = torch.zeros(512)
    xu = torch.randn(32, 80, 80, 256, generator=gg).cuda()
    def on(fn):
        def run():
            with torch.cuda.stream(mm_stream):
                for _ in range(3):
                    fn()
                mm_stream.synchronize()
        return run
    PARTNERS = [("ops.conv2d 3x3 256->256 @80^2 B=32 on a torch stream (halo tile, LDS-DMA weight ring)", on(lambda: ops.conv2d(x80, w33, None, 1, 1))),
                ("ops.linear M=204800 K=128 N=512 on a torch stream (linear tile)", on(lambda: ops.linear(xl, wl, bl))),
                ("ops.linear_ln (LayerNorm-fused linear tile)", on(lambda: ops.linear_ln(xl, wl, bl, gl, bel, 1e-6))),
                ("ops.layernorm 204800 x 128", on(lambda: ops.layernorm(xln, gl, bel, 1e-6))),
                ("ops.dwconv3x3_gelu 80^2 x 512 B=32", on(lambda: ops.dwconv3x3_gelu(xd, wd, bd))),
                ("ops.upsample2x 80^2 x 256 B=32", on(lambda: ops.upsample2x(xu)))]
if os.environ.get("PK_FORWARD"):
    PARTNERS = [(f"the forward of {os.environ['PK_FORWARD']} with {os.environ.get('PK_LABEL', 'default switches')}", forward_partner())]
FORMS = {3: "pk_fma src1 halves swapped", 4: "pk_mul src1 hi broadcast", 5: "pk_add src1 halves swapped"}
lines = []
for pname, fn in PARTNERS:
    stop = threading.Event()
    def loop():
        while not stop.is_set():
            fn()
    th = None
    if fn is not None:
        th = threading.Thread(target=loop); th.start()
    res = []
    with torch.cuda.stream(side):
        for f in FORMS:
            bad = 0
            for it in range(25):
                assert lib.pk_form_launch(f, a.data_ptr(), w.data_ptr(), out_pk.data_ptr(), out_ref.data_ptr(), BLOCKS, ITERS, side.cuda_stream) == 0
                side.synchronize()
                bad += int(not torch.equal(out_pk, out_ref))
            res.append(f"form {f}: {bad}/25")
    if th is not None:
        stop.set(); th.join()
    lines.append(f"beside {pname}: " + "  ".join(res))
    print(lines[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/pk_opsel_partners.txt", "a").write("\n".join(lines) + "\n")
