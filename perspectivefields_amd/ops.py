"""Kernel-level entry points of libpf_hip.so (pf_op_* in include/pf_hip.h) for parity tests
and tuning.  Activations are NHWC float32 torch tensors on a GPU; weights are passed in the
reference's (PyTorch) layouts and packed by the library."""
from __future__ import annotations

import ctypes

import numpy as np

from .engine import PfError, _check, _stream_ptr, load_library


def _np(a):
    if a is None:
        return None
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    return np.ascontiguousarray(a, dtype=np.float32)


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _dp(t):
    return t.data_ptr() if t is not None else None


class Planes:
    """An fp32 tensor in one of the engine's split activation formats (include/pf_hip.h): fmt "bf16x3" = three exact bf16
    planes, "f16x2" = the two fp16 planes of the split-f16 scheme.  `plane_elems` carries the format in bit 0."""

    def __init__(self, shape, device, fmt="bf16x3"):
        import torch

        self.shape = tuple(shape)
        self.fmt = fmt
        self.numel = int(np.prod(self.shape))
        stride = (self.numel + 127) // 128 * 128
        self.plane_elems = stride | (1 if fmt == "f16x2" else 0)
        self.data = torch.zeros((2 if fmt == "f16x2" else 3) * stride, dtype=torch.int16, device=device)

    def data_ptr(self):
        return self.data.data_ptr()

    def merge(self):
        """back to fp32: h + m + l (exact) or hi + lo, lo unscaled (the value the split-f16 GEMM works with)."""
        import torch

        lib = load_library()
        y = torch.empty(self.shape, dtype=torch.float32, device=self.data.device)
        _check(lib.pf_op_merge_bf16(self.data.device.index, self.data_ptr(), self.plane_elems, self.numel, y.data_ptr(), _stream_ptr()), None, "pf_op_merge_bf16")
        return y


def split_planes(x, fmt="bf16x3"):
    lib = load_library()
    x = x.contiguous()
    pl = Planes(x.shape, x.device, fmt)
    _check(lib.pf_op_split_bf16(x.device.index, x.data_ptr(), x.numel(), pl.data_ptr(), pl.plane_elems, _stream_ptr()), None, "pf_op_split_bf16")
    return pl


def _pl(planes):
    return (planes.data_ptr(), planes.plane_elems) if planes is not None else (None, 0)


def conv2d(x, weight, bias=None, stride=1, pad=0, act=0, res1=None, res2=None, post_relu=False, x2=None, nchw_out=False, tile=-1,
           planes_in=False, planes_out=False, precision=0, planes_fmt="bf16x3", splitk=True):
    """x: (B,H,W,C1) [+ x2: (B,H,W,C2) channel-concat]; weight: (Cout, C1+C2, KH, KW).  Returns (B,Ho,Wo,Cout) or NCHW.
    planes_in: hand the input(s) to the kernel as split-bf16 planes only; planes_out: take the output as planes
    (returned merged back to fp32, which is exact).  precision (split tiles): 0 split-f16 (default parity scheme), 3 exact bf16
    split, 1 bf16x3, 2 bf16.  splitk=False: never contract K in slices (the engine's split-K rule applies otherwise)."""
    import torch

    lib = load_library()
    x = x.contiguous()
    B, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    w = _np(weight)
    Cout, Cin, KH, KW = w.shape
    if Cin != C1 + C2:
        raise PfError(f"weight Cin {Cin} != {C1}+{C2}")
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    shape = (B, Cout, Ho, Wo) if nchw_out else (B, Ho, Wo, Cout)
    b = _np(bias)
    x2 = x2.contiguous() if x2 is not None else None
    xp = split_planes(x, planes_fmt) if planes_in else None
    x2p = split_planes(x2, planes_fmt) if (planes_in and x2 is not None) else None
    yp = Planes(shape, x.device, planes_fmt) if planes_out else None
    y = None if planes_out else torch.empty(shape, dtype=torch.float32, device=x.device)
    rc = lib.pf_op_conv2d(
        x.device.index, None if planes_in else x.data_ptr(), None if planes_in else _dp(x2), B, H, W, C1, C2, _hp(w), _hp(b),
        Cout, KH, KW, stride, pad, act, _dp(res1), _dp(res2), int(post_relu), int(nchw_out), tile, _dp(y),
        *_pl(xp), *_pl(x2p), *_pl(yp), precision + (0 if splitk else 16), _stream_ptr(),
    )
    _check(rc, None, "pf_op_conv2d")
    return yp.merge() if planes_out else y


def linear(x, weight, bias=None, act=0, res1=None, tile=-1):
    """x: (..., K); weight (N, K) -> (..., N)."""
    K = x.shape[-1]
    rows = x.numel() // K
    w = _np(weight)
    y = conv2d(x.reshape(1, rows, 1, K), w.reshape(w.shape[0], K, 1, 1), bias, act=act,
               res1=None if res1 is None else res1.reshape(1, rows, 1, -1), tile=tile)
    return y.reshape(*x.shape[:-1], w.shape[0])


def linear_ln(x, weight, bias, gamma, beta, eps, act=0, res1=None, tile=-1, precision=0):
    """act(Linear(LayerNorm(x))) + res1 with the LayerNorm fused into the GEMM (pf_op_linear_ln).  x: (..., K); weight (N, K)."""
    import torch

    lib = load_library()
    x = x.contiguous()
    K = x.shape[-1]
    rows = x.numel() // K
    w, b, g, be = _np(weight), _np(bias), _np(gamma), _np(beta)
    N = w.shape[0]
    y = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)
    _check(lib.pf_op_linear_ln(x.device.index, x.data_ptr(), rows, K, _hp(w), _hp(b), _hp(g), _hp(be), float(eps), N, act, _dp(res1), tile, y.data_ptr(), precision,
                               _stream_ptr()), None, "pf_op_linear_ln")
    return y


def rb_linear(x, weight, bias, tokens, gamma=None, beta=None, eps=1e-6, act=0, res=None, iters=0):
    """act(Linear(LayerNorm?(x))) + res in the row-block form (pf_op_rb_linear).  x: (rows, K) with rows = images x tokens; weight (N, K).
    iters > 0: returns the average ms per launch instead."""
    import torch

    lib = load_library()
    x = x.contiguous()
    K = x.shape[-1]
    rows = x.numel() // K
    w, b = _np(weight), _np(bias)
    g, be = (_np(gamma), _np(beta)) if gamma is not None else (None, None)
    N = w.shape[0]
    y = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)
    ms = ctypes.c_float()
    _check(lib.pf_op_rb_linear(x.device.index, x.data_ptr(), rows, tokens, K, _hp(w), _hp(b), _hp(g), _hp(be), float(eps), N,
                               act, _dp(res), y.data_ptr(), iters, ctypes.byref(ms), _stream_ptr()), None, "pf_op_rb_linear")
    return ms.value if iters > 0 else y


def rb_proj_fc1(attn, x, proj_w, proj_b, ln2_g, ln2_b, eps, fc1_w, fc1_b, tokens, iters=0):
    """x1 = x + proj(attn); hidden = fc1(LayerNorm_2(x1)) in one launch (pf_op_rb_proj_fc1).  attn, x: (rows, C) on the GPU, rows = images x tokens, C = 320.
    Returns (x1, hidden); iters > 0: the average ms per launch instead."""
    import torch

    lib = load_library()
    attn = attn.contiguous()
    x1 = x.contiguous().clone()
    rows, C = attn.shape
    hidden = torch.empty((rows, 4 * C), dtype=torch.float32, device=attn.device)
    ms = ctypes.c_float()
    a = [_np(t) for t in (proj_w, proj_b, ln2_g, ln2_b, fc1_w, fc1_b)]
    _check(lib.pf_op_rb_proj_fc1(attn.device.index, attn.data_ptr(), x1.data_ptr(), rows // tokens, tokens, C, _hp(a[0]), _hp(a[1]), _hp(a[2]), _hp(a[3]), float(eps), _hp(a[4]), _hp(a[5]),
                                 hidden.data_ptr(), iters, ctypes.byref(ms), _stream_ptr()), None, "pf_op_rb_proj_fc1")
    return ms.value if iters > 0 else (x1, hidden)


def rb_srkv(x, ln1_g, ln1_b, eps1, sr_w, sr_b, srn_g, srn_b, eps2, kv_w, kv_b, iters=0):
    """Key / value branch of a MiT block with 2 x 2 spatial reduction in one launch (pf_op_rb_srkv).  x: (B, 2 Hr, 2 Wr, C) on the GPU, C = 320.
    Returns kv (B, Hr Wr, 2 C); iters > 0: the average ms per launch instead."""
    import torch

    lib = load_library()
    x = x.contiguous()
    B, H, W, C = x.shape
    kv = torch.empty((B, (H // 2) * (W // 2), 2 * C), dtype=torch.float32, device=x.device)
    ms = ctypes.c_float()
    a = [_np(t) for t in (ln1_g, ln1_b, sr_w, sr_b, srn_g, srn_b, kv_w, kv_b)]
    _check(lib.pf_op_rb_srkv(x.device.index, x.data_ptr(), B, H // 2, W // 2, C, _hp(a[0]), _hp(a[1]), float(eps1), _hp(a[2]), _hp(a[3]), _hp(a[4]), _hp(a[5]), float(eps2),
                             _hp(a[6]), _hp(a[7]), kv.data_ptr(), iters, ctypes.byref(ms), _stream_ptr()), None, "pf_op_rb_srkv")
    return ms.value if iters > 0 else kv


def mit_attn64(x, kv, ln_gamma, ln_beta, eps, q_w, q_b, proj_w, proj_b, iters=0, inplace=False):
    """y = x + proj(softmax((LN1(x) Wq^T + bq) K^T / 8) V) for (B, N, 64) token rows and (B, M, 128) keys | values in one launch (attn_block.hip).
    iters > 0: returns (y, avg ms per launch)."""
    import torch

    lib = load_library()
    x, kv = x.contiguous(), kv.contiguous()
    B, N, C = x.shape
    y = x if inplace else torch.empty_like(x)
    ms = ctypes.c_float()
    g, b, qw, qb, pw, pb = (_np(t) for t in (ln_gamma, ln_beta, q_w, q_b, proj_w, proj_b))
    _check(lib.pf_op_mit_attn64(x.device.index, x.data_ptr(), kv.data_ptr(), y.data_ptr(), B, N, kv.shape[1], _hp(g), _hp(b), eps, _hp(qw), _hp(qb), _hp(pw), _hp(pb),
                                iters, ctypes.byref(ms), _stream_ptr()), None, "pf_op_mit_attn64")
    return (y, ms.value) if iters > 0 else y


def stem7x7(x4, weight, bias, stride, relu=False, ln_gamma=None, ln_beta=None, eps=1e-5, iters=0):
    """conv 7x7 / stride 2 or 4 / pad 3, 3 -> 64 channels (+ ReLU, or + LayerNorm over the 64 channels) on an NHWC4 image (B, H, W, 4) in one launch (stem7.hip).
    iters > 0: returns (y, avg ms per launch)."""
    import torch

    lib = load_library()
    x4 = x4.contiguous()
    B, H, W, _ = x4.shape
    Ho, Wo = (H + 6 - 7) // stride + 1, (W + 6 - 7) // stride + 1
    y = torch.empty((B, Ho, Wo, 64), dtype=torch.float32, device=x4.device)
    ms = ctypes.c_float()
    w, b = _np(weight), _np(bias)
    g, be = (_np(ln_gamma), _np(ln_beta)) if ln_gamma is not None else (None, None)
    _check(lib.pf_op_stem7x7(x4.device.index, x4.data_ptr(), y.data_ptr(), B, H, W, stride, _hp(w), _hp(b), 1 if relu else 0, _hp(g), _hp(be), float(eps), iters, ctypes.byref(ms),
                             _stream_ptr()), None, "pf_op_stem7x7")
    return (y, ms.value) if iters > 0 else y


def thin128(x, weight, bias, res=None, iters=0, inplace=False):
    """x W^T + b (+ res) for a 128 -> 128 layer over (..., 128) rows in the transposed, register-epilogue form (thin_linear.hip).  inplace: y aliases res.
    iters > 0: returns (y, avg ms per launch)."""
    import torch

    lib = load_library()
    x = x.contiguous()
    rows = x.numel() // 128
    y = res if (inplace and res is not None) else torch.empty_like(x)
    ms = ctypes.c_float()
    w, b = _np(weight), _np(bias)
    _check(lib.pf_op_thin128(x.device.index, x.data_ptr(), rows, _hp(w), _hp(b), _dp(res), y.data_ptr(), iters, ctypes.byref(ms), _stream_ptr()), None, "pf_op_thin128")
    return (y, ms.value) if iters > 0 else y


def mit_mlp(x, fc1_w, fc1_b, ln_gamma, ln_beta, eps, dw_w, dw_b, fc2_w, fc2_b, iters=0):
    """One MiT block Mlp in one kernel: x + fc2(GELU(dwconv3x3(fc1(LayerNorm(x))))).  x: (B, Hs, Ws, C) on the GPU, C = 64 or 128.
    iters > 0: returns the average ms per launch instead."""
    import torch

    lib = load_library()
    x = x.contiguous()
    B, Hs, Ws, C = x.shape
    y = torch.empty_like(x)
    ms = ctypes.c_float()
    a = [_np(v) for v in (fc1_w, fc1_b, ln_gamma, ln_beta, dw_w, dw_b, fc2_w, fc2_b)]
    _check(lib.pf_op_mit_mlp(x.device.index, x.data_ptr(), y.data_ptr(), B, Hs, Ws, C, _hp(a[0]), _hp(a[1]), _hp(a[2]), _hp(a[3]), float(eps), _hp(a[4]), _hp(a[5]), _hp(a[6]), _hp(a[7]),
                             iters, ctypes.byref(ms), _stream_ptr()), None, "pf_op_mit_mlp")
    return ms.value if iters > 0 else y


def cnx_mlp(d, y, w1, b1, ln_gamma, ln_beta, eps, w2, b2, layer_scale, iters=0):
    """One ConvNeXt block MLP in one kernel: returns y + layer_scale * pwconv2(GELU(pwconv1(LayerNorm(d)))) (y is not modified: a copy is
    updated).  d, y: (rows, C) on the GPU, C = 96 or 192.  iters > 0: returns (result of the first launch is lost) the average ms per launch."""
    lib = load_library()
    d = d.contiguous()
    out = y.contiguous().clone()
    rows, C = d.shape
    ms = ctypes.c_float()
    a = [_np(v) for v in (w1, b1, ln_gamma, ln_beta, w2, b2, layer_scale)]
    _check(lib.pf_op_cnx_mlp(d.device.index, d.data_ptr(), out.data_ptr(), rows, C, _hp(a[0]), _hp(a[1]), _hp(a[2]), _hp(a[3]), float(eps), _hp(a[4]), _hp(a[5]), _hp(a[6]),
                             iters, ctypes.byref(ms), _stream_ptr()), None, "pf_op_cnx_mlp")
    return ms.value if iters > 0 else out


def layernorm(x, gamma, beta, eps, planes_out=False):
    import torch

    lib = load_library()
    x = x.contiguous()
    C = x.shape[-1]
    y = None if planes_out else torch.empty_like(x)
    yp = Planes(x.shape, x.device) if planes_out else None
    g, b = _np(gamma), _np(beta)
    _check(lib.pf_op_layernorm(x.device.index, x.data_ptr(), _hp(g), _hp(b), _dp(y), x.numel() // C, C, eps, *_pl(yp), _stream_ptr()), None, "pf_op_layernorm")
    return yp.merge() if planes_out else y


def dwconv3x3_gelu(x, weight, bias, planes_out=False, variant=None):
    """variant None: the default path; otherwise an explicit kernel (pf_op_dwconv3x3_bench's ids)."""
    import torch

    lib = load_library()
    x = x.contiguous()
    B, H, W, C = x.shape
    y = None if planes_out else torch.empty_like(x)
    yp = Planes(x.shape, x.device) if planes_out else None
    w, b = _np(weight), _np(bias)
    if variant is not None:
        _check(lib.pf_op_dwconv3x3_gelu_cfg(x.device.index, x.data_ptr(), _hp(w), _hp(b), _dp(y), B, H, W, C, *_pl(yp), int(variant), _stream_ptr()), None, "pf_op_dwconv3x3_gelu_cfg")
        return yp.merge() if planes_out else y
    _check(lib.pf_op_dwconv3x3_gelu(x.device.index, x.data_ptr(), _hp(w), _hp(b), _dp(y), B, H, W, C, *_pl(yp), _stream_ptr()), None, "pf_op_dwconv3x3_gelu")
    return yp.merge() if planes_out else y


def dwconv7x7(x, weight, bias, variant=None, nc=0, nb=0, th=0):
    """variant None: the default path; 3: column-blocked streaming kernel (nc columns per thread, nb row buffers, strips of th
    rows; 0 = automatic); 2: one column per lane."""
    import torch

    lib = load_library()
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    w, b = _np(weight), _np(bias)
    if variant is None:
        _check(lib.pf_op_dwconv7x7(x.device.index, x.data_ptr(), _hp(w), _hp(b), y.data_ptr(), B, H, W, C, _stream_ptr()), None, "pf_op_dwconv7x7")
    else:
        _check(lib.pf_op_dwconv7x7_cfg(x.device.index, x.data_ptr(), _hp(w), _hp(b), y.data_ptr(), B, H, W, C, variant, nc, nb, th, _stream_ptr()), None, "pf_op_dwconv7x7_cfg")
    return y


def dwconv7x7_bench(variant, B, H, W, C, nc=0, nb=0, th=0, iters=10, device=0):
    lib = load_library()
    ms = ctypes.c_float()
    _check(lib.pf_op_dwconv7x7_bench(device, variant, nc, nb, th, B, H, W, C, iters, ctypes.byref(ms)), None, "pf_op_dwconv7x7_bench")
    return ms.value


def sr_attention(q, kv, heads, planes_out=False):
    """q: (B,N,C), kv: (B,M,2C) -> (B,N,C); head_dim 64."""
    import torch

    lib = load_library()
    q, kv = q.contiguous(), kv.contiguous()
    B, N, C = q.shape
    M = kv.shape[1]
    out = None if planes_out else torch.empty_like(q)
    op = Planes(q.shape, q.device) if planes_out else None
    _check(lib.pf_op_sr_attention(q.device.index, q.data_ptr(), kv.data_ptr(), _dp(out), B, N, M, heads, *_pl(op), _stream_ptr()), None, "pf_op_sr_attention")
    return op.merge() if planes_out else out


def sr_attention_variant(q, kv, heads, variant, iters=0):
    """variant 1: split-f16 MFMA kernel (the forward's default), 0: exact fp32 MFMA.  iters > 0: returns (out, avg ms per launch)."""
    import torch

    lib = load_library()
    q, kv = q.contiguous(), kv.contiguous()
    B, N, C = q.shape
    out = torch.empty_like(q)
    ms = ctypes.c_float()
    _check(lib.pf_op_sr_attention_variant(q.device.index, variant, q.data_ptr(), kv.data_ptr(), out.data_ptr(), B, N, kv.shape[1], heads, iters,
                                          ctypes.byref(ms), _stream_ptr()), None, "pf_op_sr_attention_variant")
    return (out, ms.value) if iters > 0 else out


def upsample2x(x, planes_out=False):
    import torch

    lib = load_library()
    x = x.contiguous()
    B, H, W, C = x.shape
    shape = (B, 2 * H, 2 * W, C)
    y = None if planes_out else torch.empty(shape, dtype=torch.float32, device=x.device)
    yp = Planes(shape, x.device) if planes_out else None
    _check(lib.pf_op_upsample2x(x.device.index, x.data_ptr(), _dp(y), B, H, W, C, *_pl(yp), _stream_ptr()), None, "pf_op_upsample2x")
    return yp.merge() if planes_out else y


def conv_tiles():
    lib = load_library()
    return [lib.pf_op_conv_tile_name(i).decode() for i in range(lib.pf_op_num_conv_tiles())]


def conv2d_bench(B, H, W, Cin, Cout, K, stride=1, pad=0, tile=-1, iters=10, device=0, fmt=0, precision=0):
    """Average ms per launch of one conv shape on random data (tuning aid).  fmt 0: fp32 in / out; 1: split-plane
    input; 2: split-plane input and output (the exact bf16 split).  precision: PF_PRECISION_* of the split tiles
    (0 split-f16 default, 3 exact bf16 split, 1 bf16x3, 2 bf16).  Returns -1 when the tile cannot run the format."""
    lib = load_library()
    ms = ctypes.c_float()
    _check(lib.pf_op_conv2d_bench(device, B, H, W, Cin, Cout, K, stride, pad, tile, iters, fmt + 16 * precision, ctypes.byref(ms)), None, "pf_op_conv2d_bench")
    return ms.value


def dwconv3x3_bench(variant, B, H, W, C, iters=10, device=0):
    lib = load_library()
    ms = ctypes.c_float()
    _check(lib.pf_op_dwconv3x3_bench(device, variant, B, H, W, C, iters, ctypes.byref(ms)), None, "pf_op_dwconv3x3_bench")
    return ms.value
