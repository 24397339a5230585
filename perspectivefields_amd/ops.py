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


def conv2d(x, weight, bias=None, stride=1, pad=0, act=0, res1=None, res2=None, post_relu=False, x2=None, nchw_out=False, tile=-1):
    """x: (B,H,W,C1) [+ x2: (B,H,W,C2) channel-concat]; weight: (Cout, C1+C2, KH, KW).  Returns (B,Ho,Wo,Cout) or NCHW."""
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
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    b = _np(bias)
    rc = lib.pf_op_conv2d(
        x.device.index, x.data_ptr(), _dp(x2.contiguous() if x2 is not None else None), B, H, W, C1, C2, _hp(w), _hp(b),
        Cout, KH, KW, stride, pad, act, _dp(res1), _dp(res2), int(post_relu), int(nchw_out), tile, y.data_ptr(), _stream_ptr(),
    )
    _check(rc, None, "pf_op_conv2d")
    return y


def linear(x, weight, bias=None, act=0, res1=None, tile=-1):
    """x: (..., K); weight (N, K) -> (..., N)."""
    K = x.shape[-1]
    rows = x.numel() // K
    w = _np(weight)
    y = conv2d(x.reshape(1, rows, 1, K), w.reshape(w.shape[0], K, 1, 1), bias, act=act,
               res1=None if res1 is None else res1.reshape(1, rows, 1, -1), tile=tile)
    return y.reshape(*x.shape[:-1], w.shape[0])


def layernorm(x, gamma, beta, eps):
    import torch

    lib = load_library()
    x = x.contiguous()
    C = x.shape[-1]
    y = torch.empty_like(x)
    g, b = _np(gamma), _np(beta)
    _check(lib.pf_op_layernorm(x.device.index, x.data_ptr(), _hp(g), _hp(b), y.data_ptr(), x.numel() // C, C, eps, _stream_ptr()), None, "pf_op_layernorm")
    return y


def dwconv3x3_gelu(x, weight, bias):
    import torch

    lib = load_library()
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    w, b = _np(weight), _np(bias)
    _check(lib.pf_op_dwconv3x3_gelu(x.device.index, x.data_ptr(), _hp(w), _hp(b), y.data_ptr(), B, H, W, C, _stream_ptr()), None, "pf_op_dwconv3x3_gelu")
    return y


def dwconv7x7(x, weight, bias):
    import torch

    lib = load_library()
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    w, b = _np(weight), _np(bias)
    _check(lib.pf_op_dwconv7x7(x.device.index, x.data_ptr(), _hp(w), _hp(b), y.data_ptr(), B, H, W, C, _stream_ptr()), None, "pf_op_dwconv7x7")
    return y


def sr_attention(q, kv, heads):
    """q: (B,N,C), kv: (B,M,2C) -> (B,N,C); head_dim 64."""
    import torch

    lib = load_library()
    q, kv = q.contiguous(), kv.contiguous()
    B, N, C = q.shape
    M = kv.shape[1]
    out = torch.empty_like(q)
    _check(lib.pf_op_sr_attention(q.device.index, q.data_ptr(), kv.data_ptr(), out.data_ptr(), B, N, M, heads, _stream_ptr()), None, "pf_op_sr_attention")
    return out


def upsample2x(x):
    import torch

    lib = load_library()
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty((B, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    _check(lib.pf_op_upsample2x(x.device.index, x.data_ptr(), y.data_ptr(), B, H, W, C, _stream_ptr()), None, "pf_op_upsample2x")
    return y


def conv_tiles():
    lib = load_library()
    return [lib.pf_op_conv_tile_name(i).decode() for i in range(lib.pf_op_num_conv_tiles())]


def conv2d_bench(B, H, W, Cin, Cout, K, stride=1, pad=0, tile=-1, iters=10, device=0):
    """Average ms per launch of one conv shape on random data (tuning aid)."""
    lib = load_library()
    ms = ctypes.c_float()
    _check(lib.pf_op_conv2d_bench(device, B, H, W, Cin, Cout, K, stride, pad, tile, iters, ctypes.byref(ms)), None, "pf_op_conv2d_bench")
    return ms.value


def dwconv3x3_bench(variant, B, H, W, C, iters=10, device=0):
    lib = load_library()
    ms = ctypes.c_float()
    _check(lib.pf_op_dwconv3x3_bench(device, variant, B, H, W, C, iters, ctypes.byref(ms)), None, "pf_op_dwconv3x3_bench")
    return ms.value
