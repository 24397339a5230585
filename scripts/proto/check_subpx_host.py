"""CPU check of the HOST side of the sub-pixel conv (tuning build: PF_TUNING_BUILD=1 at build and run time): pf_tuning_subpx_combine's weights / tables, used with a
numpy mirror of the device code's indexing (subpx_corr_kernel in elem.hip, epilogue_subpx in igemm_common.h), against torch's interpolate + conv2d in float64."""
import ctypes, os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from perspectivefields_amd.engine import load_library

lib = load_library()
fn = getattr(lib, "pf_tuning_subpx_combine", None)
if fn is None:
    raise SystemExit("product build: build and run with PF_TUNING_BUILD=1")
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


def run(B, H, W, Cin, Cr, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)  # NHWC like the device tensor
    w = (rng.standard_normal((Cr, Cin, 3, 3)) * 0.1).astype(np.float32)
    weff = np.zeros((4 * Cr, Cin, 3, 3), np.float32)
    S = Cin * Cr
    tab = np.zeros(28 * S, np.float32)
    assert fn(w.ctypes.data, Cr, Cin, weff.ctypes.data, tab.ctypes.data) == 0
    top, bot, lef, rig, cor = (tab[k * S:(k + n) * S] for k, n in ((0, 6), (6, 6), (12, 6), (18, 6), (24, 4)))
    xd = x.astype(np.float64)
    # main contraction: 3x3 conv on the replicate-padded half-resolution map, 4 Cr virtual channels
    xp = np.pad(xd, ((0, 0), (1, 1), (1, 1), (0, 0)), mode="edge")
    acc = np.zeros((B, H, W, 4 * Cr))
    for dy in range(3):
        for dx in range(3):
            acc += np.einsum("bhwi,ni->bhwn", xp[:, dy:dy + H, dx:dx + W, :], weff[:, :, dy, dx].astype(np.float64))
    # subpx_corr_kernel mirrored: perimeter index pi -> (oy, ox); thread n -> (py, px, c)
    PER = 2 * W + 2 * H
    corr = np.zeros((B, PER, 4 * Cr))
    X = lambda b, iy, ix: xd[b, min(max(iy, 0), H - 1), min(max(ix, 0), W - 1), :]
    for b in range(B):
        for pi in range(PER):
            if pi < W: oy, ox = 0, pi
            elif pi < 2 * W: oy, ox = H - 1, pi - W
            elif pi < 2 * W + H: oy, ox = pi - 2 * W, 0
            else: oy, ox = pi - 2 * W - H, W - 1
            for n in range(4 * Cr):
                phase, c = divmod(n, Cr); py, px = phase >> 1, phase & 1
                t, bt, l, r = oy == 0 and py == 0, oy == H - 1 and py == 1, ox == 0 and px == 0, ox == W - 1 and px == 1
                a = 0.0
                if t or bt:
                    tb = (top if t else bot)
                    for dx in range(3):
                        xr = X(b, 0 if t else H - 1, ox + dx - 1)
                        a -= sum(float(tb[(px * 3 * S) + c + (dx * Cin + ci) * Cr]) * xr[ci] for ci in range(Cin))
                if l or r:
                    tb = (lef if l else rig)
                    for dy in range(3):
                        xr = X(b, oy + dy - 1, 0 if l else W - 1)
                        a -= sum(float(tb[(py * 3 * S) + c + (dy * Cin + ci) * Cr]) * xr[ci] for ci in range(Cin))
                if (t or bt) and (l or r):
                    xr = X(b, oy, ox)
                    a += sum(float(cor[(py * 2 + px) * S + c + ci * Cr]) * xr[ci] for ci in range(Cin))
                corr[b, pi, n] = a
    # epilogue_subpx mirrored: border pixels add corr[pi(oy, ox)], then the pixel shuffle
    y = np.zeros((B, 2 * H, 2 * W, Cr))
    for oy in range(H):
        for ox in range(W):
            v = acc[:, oy, ox, :].copy()
            if oy == 0 or oy == H - 1 or ox == 0 or ox == W - 1:
                pi = ox if oy == 0 else (W + ox if oy == H - 1 else (2 * W + oy if ox == 0 else 2 * W + H + oy))
                v += corr[:, pi, :]
            for phase in range(4):
                y[:, 2 * oy + (phase >> 1), 2 * ox + (phase & 1), :] = v[:, phase * Cr:(phase + 1) * Cr]
    up = F.interpolate(torch.from_numpy(xd).permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    ref = F.conv2d(up, torch.from_numpy(w.astype(np.float64)), padding=1).permute(0, 2, 3, 1).numpy()
    err = np.abs(y - ref).max()
    print(f"B{B} {H}x{W} Cin{Cin} Cr{Cr}: max |err| {err:.2e} (fp32-rounded combined weights)")
    assert err < 1e-5 * max(1.0, np.abs(ref).max())


for args in [(1, 3, 4, 4, 2, 0), (2, 1, 1, 3, 2, 1), (1, 2, 1, 2, 3, 2), (1, 5, 6, 8, 4, 3)]:
    run(*args)
print("ok")
