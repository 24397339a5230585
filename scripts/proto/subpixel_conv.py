"""Algebra check (CPU, numpy / torch) for the planned "sub-pixel" form of the decoders' fused up-sampling convs
(reference: decode_head.py:284-286 / gravity_head.py:172 -- F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) followed by a 3x3 / pad 1 conv).

conv3x3(up2(x)) at output pixel (2i + py, 2j + px) is a 3x3 conv on the HALF-resolution map around (i, j) with phase-specific, host-combined weights
    W_eff[py][px][dy][dx] = sum_{ky, kx} A[py][dy][ky] * A[px][dx][kx] * w[ky][kx]
when x is extended by REPLICATION (x[-1] = x[0], x[H] = x[H-1]: the interpolation's clamping).  What that gets wrong is only the conv's ZERO padding of the
up-sampled map: taps that land on u[-1] / u[2H] (output rows 0 and 2H - 1, columns 0 and 2W - 1) see the replicated value instead of 0.  Those taps have a closed form
too -- u_hat[-1] = x[0] -- so border pixels just use other effective weights (delta_* below), i.e. a correction product with the rows of non-border pixels masked.

Run: python scripts/proto/subpixel_conv.py   (asserts max |error| ~1e-6 against torch's interpolate + conv2d in float64)"""
import numpy as np
import torch
import torch.nn.functional as F

# A[p][d + 1][k]: coefficient of half-resolution neighbour d (-1, 0, +1) in tap k (0, 1, 2) of output phase p, one dimension
A = np.zeros((2, 3, 3))
A[0] = [[0.75, 0.25, 0.0], [0.25, 0.75, 0.75], [0.0, 0.0, 0.25]]
A[1] = [[0.25, 0.0, 0.0], [0.75, 0.75, 0.25], [0.0, 0.25, 0.75]]


def effective_weights(w):
    """w: (Cout, Cin, 3, 3) -> W_eff (2, 2, Cout, Cin, 3, 3): [py][px][..][dy][dx]"""
    return np.einsum("pyk,qxl,oikl->pqoiyx", A, A, w)


def border_deltas(w):
    """Effective weights of the taps that land on the zero padding, to be SUBTRACTED for border pixels.
    Top row (output row 0 = phase py 0 of half-res row 0): tap ky = 0 reads u_hat[-1, X] = interpolation of x[0, :] along x -> a 1x3 half-res conv per px:
      top[px][dx] = sum_kx A[px][dx][kx] w[0][kx]          (applied to x[0, j + dx - 1])
    bottom (output row 2H - 1 = phase py 1 of half-res row H - 1): ky = 2 likewise on x[H - 1, :]; left / right: the same with the roles of y and x exchanged.
    Corners: both a row tap and a column tap are padding: inclusion-exclusion -- the (ky, kx) = corner tap was subtracted twice, add it back once:
      it reads u_hat[-1, -1] = x[0, 0] (coefficient 1)."""
    top = np.einsum("qxl,oil->qoix", A, w[:, :, 0, :])     # [px][o][i][dx]
    bottom = np.einsum("qxl,oil->qoix", A, w[:, :, 2, :])
    left = np.einsum("pyk,oik->poiy", A, w[:, :, :, 0])    # [py][o][i][dy]
    right = np.einsum("pyk,oik->poiy", A, w[:, :, :, 2])
    return top, bottom, left, right


def subpixel_conv(x, w):
    """x: (Cin, H, W) float64, w: (Cout, Cin, 3, 3) -> (Cout, 2H, 2W) = conv3x3_pad1(bilinear_up2(x)) without ever forming the up-sampled map"""
    Cin, H, W = x.shape
    Cout = w.shape[0]
    We = effective_weights(w)
    top, bottom, left, right = border_deltas(w)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)), mode="edge")  # replication = the interpolation's clamping
    y = np.zeros((Cout, 2 * H, 2 * W))
    for py in range(2):
        for px in range(2):
            acc = np.zeros((Cout, H, W))
            for dy in range(3):
                for dx in range(3):
                    acc += np.einsum("oi,ihw->ohw", We[py, px, :, :, dy, dx], xp[:, dy:dy + H, dx:dx + W])
            # ---- border corrections (only pixels on the image border; everything else above is the plain phase conv)
            if py == 0:  # output row 0: half-res row 0
                for dx in range(3):
                    acc[:, 0, :] -= np.einsum("oi,iw->ow", top[px][:, :, dx], xp[:, 1, dx:dx + W])
            else:        # output row 2H - 1: half-res row H - 1
                for dx in range(3):
                    acc[:, H - 1, :] -= np.einsum("oi,iw->ow", bottom[px][:, :, dx], xp[:, H, dx:dx + W])
            if px == 0:
                for dy in range(3):
                    acc[:, :, 0] -= np.einsum("oi,ih->oh", left[py][:, :, dy], xp[:, dy:dy + H, 1])
            else:
                for dy in range(3):
                    acc[:, :, W - 1] -= np.einsum("oi,ih->oh", right[py][:, :, dy], xp[:, dy:dy + H, W])
            # corners: the corner tap was subtracted by the row AND the column correction
            ci, cj = (0 if py == 0 else H - 1), (0 if px == 0 else W - 1)
            ky, kx = (0 if py == 0 else 2), (0 if px == 0 else 2)
            acc[:, ci, cj] += w[:, :, ky, kx] @ x[:, ci, cj]
            y[:, py::2, px::2] = acc
    return y


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for (cin, cout, h, w_) in [(5, 4, 6, 7), (3, 2, 1, 1), (4, 3, 2, 5), (2, 2, 1, 4), (8, 8, 9, 3)]:
        x = rng.standard_normal((cin, h, w_))
        w = rng.standard_normal((cout, cin, 3, 3))
        ref = F.conv2d(F.interpolate(torch.from_numpy(x)[None], scale_factor=2, mode="bilinear", align_corners=False), torch.from_numpy(w), padding=1)[0].numpy()
        got = subpixel_conv(x, w)
        err = np.abs(got - ref).max()
        print(f"Cin {cin} Cout {cout} {h}x{w_}: max |err| {err:.2e}")
        assert err < 1e-9, err
    print("ok")
