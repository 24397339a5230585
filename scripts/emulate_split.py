"""CPU emulation of a reduced-term split contraction across the WHOLE network (decision aid, test infrastructure only).

Question: can the dense contractions run on fewer MFMAs than the 6-term exact bf16 split and stay
inside the parity tolerances (up-vector 1-cos / latitude L1 1e-3, ParamNet scalars 1e-4) with >= 3x margin?

Candidates:
  f16x3  : operands as hi + lo * 2^-11, hi = fp16_rn(x), lo = fp16_rn((x - hi) * 2^11); products hi*hi + hi*lo + lo*hi
           (3 x v_mfma_f32_32x32x16_f16).  Weights are scaled per output channel by a power of two so that the row
           maximum sits in [2^13, 2^14) (exact; undone in the epilogue).
  bf16x3 : the existing reduced mode (h, m of the bf16 truncation split; 3 products).

The emulation replaces every dense conv / linear of the oracle whose input channel count is a multiple of 32 (the
layers the split kernels serve) by: quantise both operands to what the scheme represents, contract in fp64 (the MFMA
partial products are exact in fp32; accumulation noise is modelled by the plain fp32 oracle), drop the lo*lo term,
round the result to fp32.  Everything else is the fp32 oracle.  Reported: scalar / field error of (a) the plain fp32
oracle and (b) the emulated scheme against the float64 run of the oracle on the same inputs.

    python scripts/emulate_split.py            # run from the repo root, CPU only (~2 min)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pf_oracle  # noqa: E402
from perspectivefields_amd.config import arch_of, get_cfg  # noqa: E402
from perspectivefields_amd.synth import synthetic_image, synthetic_state_dict, to_torch  # noqa: E402

_conv2d, _linear = F.conv2d, F.linear
SCHEME = "f16x3"


def split_f16(x64):
    """x (float64 holding fp32 values) -> (hi, lo_true) as float64: hi = fp16_rn(x), lo_true = fp16_rn((x-hi)*2^11)/2^11."""
    x32 = x64.to(torch.float32)
    hi = x32.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)
    lo = ((x32 - hi) * 2048.0).to(torch.float16).to(torch.float32)
    return hi.double(), lo.double() / 2048.0


def split_f16_unscaled(x64, flush):
    """lo kept UNSCALED: lo = fp16_rn(x - hi) -- fp16 subnormals below 2^-14 (|x| < ~0.25), preserved or flushed to zero.  Would save the wh 2^-11 operand
    (4 v_pk_mul_f16 per weight fragment in every split kernel) if the matrix cores keep fp16 subnormal inputs."""
    x32 = x64.to(torch.float32)
    hi = x32.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)
    lo = (x32 - hi).to(torch.float16).to(torch.float32)
    if flush:
        lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
    return hi.double(), lo.double()


def split_bf16_hm(x64):
    x32 = x64.to(torch.float32)
    h = (x32.view(torch.int32) & -65536).view(torch.float32)
    m = (x32 - h).to(torch.bfloat16).to(torch.float32)
    return h.double(), m.double()


def weight_scale(w):
    """per-output-channel power of two bringing max|w| into [2^13, 2^14)"""
    mx = w.abs().flatten(1).max(dim=1).values.double()
    e = torch.where(mx > 0, 13.0 - torch.floor(torch.log2(mx.clamp_min(1e-300))), torch.zeros_like(mx))
    return torch.pow(torch.tensor(2.0, dtype=torch.float64), e)


RED = ""  # reduced scheme of the decoders' 3x3 / stride-1 convs ("mix_*" modes): a1w2 = ah (wh + wl), a2w1 = (ah + al) wh, a1w1 = ah wh; everything else stays SCHEME


def contract(op, x, w, **kw):
    xd, wd = x.double(), w.double()
    if RED and op is _conv2d and tuple(w.shape[2:]) == (3, 3) and kw.get("stride", 1) in (1, (1, 1)):
        s = weight_scale(w)
        xh, xl = split_f16_unscaled(xd, False)
        wh, wl = split_f16(wd * s.view(-1, 1, 1, 1))
        y = op(xh, wh, **kw)
        if RED == "a1w2":
            y = y + op(xh, wl, **kw)
        elif RED == "a2w1":
            y = y + op(xl, wh, **kw)
        return (y / s.view(1, -1, 1, 1)).to(torch.float32)
    if SCHEME in ("f16x3", "f16x3u", "f16x3uf"):
        s = weight_scale(w)
        sh = [-1] + [1] * (w.dim() - 1)
        xh, xl = split_f16(xd) if SCHEME == "f16x3" else split_f16_unscaled(xd, SCHEME == "f16x3uf")
        wh, wl = split_f16(wd * s.view(sh))
        y = op(xh, wh, **kw) + op(xh, wl, **kw) + op(xl, wh, **kw)
        shape = [1, -1, 1, 1] if op is _conv2d else [-1]
        y = y / s.view(shape)
    else:
        xh, xm = split_bf16_hm(xd)
        wh, wm = split_bf16_hm(wd)
        y = op(xh, wh, **kw) + op(xh, wm, **kw) + op(xm, wh, **kw)
    return y.to(torch.float32)


WINOGRAD = False
_BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
_G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
_AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def conv3x3_winograd(x, w, b):
    """Winograd F(2x2, 3x3) as a kernel would run it: input transform B^T d B in fp32 (adds only), weight transform G g G^T in fp64 at
    finalize, the 16 position GEMMs on the split-f16 scheme (weights scaled per (position, output channel) by a power of two; fp64 accumulation stands for the fp32 MFMA
    accumulator as elsewhere in this file), output transform A^T m A in fp32."""
    B, C, H, W = x.shape
    O = w.shape[0]
    He, We = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + We - W, 1, 1 + He - H))                                      # zero padding of the conv + round up to whole 2x2 output tiles
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                              # (B, C, ty, tx, 4, 4) fp32
    bt = _BT.to(torch.float32)
    v = torch.einsum("ij,bcyxjk->bcyxik", bt, d)                                        # fp32 adds (entries of B are 0, +-1)
    v = torch.einsum("bcyxik,lk->bcyxil", v, bt)
    u = torch.einsum("ij,ocjk,lk->ocil", _G, w.double(), _G)                            # (O, C, 4, 4) fp64
    ty, tx = v.shape[2], v.shape[3]
    m = torch.empty((B, O, ty, tx, 4, 4), dtype=torch.float32)
    for i in range(4):
        for j in range(4):
            uw = u[:, :, i, j]                                                          # (O, C)
            s = weight_scale(uw)
            wh, wl = split_f16(uw * s.view(-1, 1))
            ah, al = split_f16(v[..., i, j].double()) if SCHEME == "f16x3" else split_f16_unscaled(v[..., i, j].double(), False)
            y = torch.einsum("bcyx,oc->boyx", ah, wh) + torch.einsum("bcyx,oc->boyx", ah, wl) + torch.einsum("bcyx,oc->boyx", al, wh)
            m[..., i, j] = (y / s.view(1, -1, 1, 1)).to(torch.float32)
    at = _AT.to(torch.float32)
    o = torch.einsum("ij,boyxjk->boyxik", at, m)
    o = torch.einsum("boyxik,lk->boyxil", o, at)                                        # (B, O, ty, tx, 2, 2) fp32
    o = o.permute(0, 1, 2, 4, 3, 5).reshape(B, O, 2 * ty, 2 * tx)[:, :, :H, :W]
    return o if b is None else o + b.view(1, -1, 1, 1)


def conv2d_q(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if groups != 1 or x.shape[1] % 32 != 0 or x.dtype != torch.float32:
        return _conv2d(x, w, b, stride, padding, dilation, groups)
    if WINOGRAD and tuple(w.shape) == (256, 256, 3, 3) and stride in (1, (1, 1)) and padding in (1, (1, 1)):
        return conv3x3_winograd(x, w, b)
    y = contract(_conv2d, x, w, stride=stride, padding=padding)
    return y if b is None else y + b.view(1, -1, 1, 1)


def linear_q(x, w, b=None):
    if x.shape[-1] % 32 != 0 or x.dtype != torch.float32:
        return _linear(x, w, b)
    y = contract(_linear, x, w)
    return y if b is None else y + b


def attention_q(w, x, H, W, heads, sr):
    """mit_attention with both attention products on the split-f16 scheme (K and V scaled by 64 and split as hi + lo_true,
    q * d^-0.5 and P split as hi + lo 2^-11): the planned MFMA form of csrc/attn.hip."""
    B, N, C = x.shape
    d = C // heads
    q = F.linear(x, w("q.weight"), w("q.bias")).reshape(B, N, heads, d).transpose(1, 2)
    if sr > 1:
        xm = x.transpose(1, 2).reshape(B, C, H, W)
        xr = F.conv2d(xm, w("sr.weight"), w("sr.bias"), stride=sr)
        xr = xr.reshape(B, C, -1).transpose(1, 2)
        xr = pf_oracle.layer_norm(xr, w("norm.weight"), w("norm.bias"), 1e-5)
    else:
        xr = x
    kv = F.linear(xr, w("kv.weight"), w("kv.bias")).reshape(B, -1, 2, heads, d)
    k = kv[:, :, 0].transpose(1, 2)
    v = kv[:, :, 1].transpose(1, 2)

    def mm(a, b_scaled64):  # a: activations split hi + lo/2048; b: (64 b) split hi + lo_true; the lo*lo term is dropped
        ah, al = split_f16(a.double()) if SCHEME == "f16x3" else split_f16_unscaled(a.double(), SCHEME == "f16x3uf")
        bh, bl = split_f16(b_scaled64.double() * 64.0)
        return ((ah @ bh + ah @ bl + al @ bh) / 64.0).to(torch.float32)

    a = mm(q * (d ** -0.5), k.transpose(-2, -1))
    a = a.softmax(dim=-1)
    o = mm(a, v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, w("proj.weight"), w("proj.bias"))


_attention = pf_oracle.mit_attention


def run(sd, arch, imgs, mode):
    global SCHEME, WINOGRAD, RED
    RED = ""
    if mode.startswith("mix_"):
        RED, mode = mode[4:], "f16x3u+attn"
    WINOGRAD = mode.endswith("+wino")
    if WINOGRAD:
        mode = mode[:-5]
    if mode in ("fp32", "fp64"):
        F.conv2d, F.linear = _conv2d, _linear
        dtype = torch.float64 if mode == "fp64" else torch.float32
    else:
        SCHEME = "f16x3uf" if mode.startswith("f16x3uf") else ("f16x3u" if mode.startswith("f16x3u") else ("f16x3" if mode.startswith("f16x3") else mode))
        F.conv2d, F.linear = conv2d_q, linear_q
        if mode.endswith("+attn"):
            pf_oracle.mit_attention = attention_q
        dtype = torch.float32
    try:
        with torch.no_grad():
            return pf_oracle.inference_batch(sd, arch, imgs, dtype)
    finally:
        F.conv2d, F.linear = _conv2d, _linear
        pf_oracle.mit_attention = _attention


def main():
    version = "Paramnet-360Cities-edina-centered"
    sd = to_torch(synthetic_state_dict(version, 0))
    arch = arch_of(get_cfg(version))
    torch.set_num_threads(os.cpu_count() or 1)
    imgs = [synthetic_image(640, 640, seed=1000 + i) for i in range(int(os.environ.get("N_IMG", "3")))]
    keys = ("pred_roll", "pred_pitch", "pred_vfov", "pred_rel_focal")
    truth = run(sd, arch, imgs, "fp64")
    for mode in os.environ.get("MODES", "fp32,f16x3,f16x3+attn,bf16x3").split(","):
        res = run(sd, arch, imgs, mode)
        dpar, dcos, dlat = 0.0, 0.0, 0.0
        for r, t in zip(res, truth):
            dpar = max(dpar, max(abs(float(r[k]) - float(t[k])) for k in keys))
            g, go = r["pred_gravity_original"].double(), t["pred_gravity_original"].double()
            c = 1.0 - (g * go).sum(0) / torch.sqrt((g * g).sum(0) * (go * go).sum(0))
            dcos = max(dcos, float(c.max()))
            dlat = max(dlat, float((r["pred_latitude_original"].double() - t["pred_latitude_original"].double()).abs().mean()))
        print(f"{mode:10s} vs fp64: ParamNet max|d| {dpar:.3e} (tol 1e-4)   up 1-cos max {dcos:.3e} (tol 1e-3)   latitude L1 {dlat:.3e} deg (tol 1e-3)", flush=True)


if __name__ == "__main__":
    main()
