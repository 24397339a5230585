"""Kernel-level parity (-m gpu): every HIP kernel of libpf_hip.so, called through the C ABI
(pf_op_*), against the CPU oracle's formulation of the same op in float64.
Tolerance: fp32 accumulation error, |err| <= 2e-5 * (1 + |ref|) * sqrt(K/64)-ish; stated per test."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pf_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from perspectivefields_amd import ops as _ops

    return _ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale)


def _close(got, ref, tol, what):
    got = got.double().cpu()
    ref = ref.double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs()
    bound = tol * (1.0 + ref.abs())
    worst = float((err / bound).max())
    assert worst <= 1.0, f"{what}: max err {float(err.max()):.3e} (ratio to bound {worst:.2f}), ref scale {float(ref.abs().max()):.3e}"


def _ref_conv(x_nhwc, w, b, stride, pad, x2=None):
    x = x_nhwc if x2 is None else torch.cat([x_nhwc, x2], dim=-1)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None if b is None else b.double(), stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # name, B, H, W, C1, C2, Cout, K, stride, pad
    ("head3x3_64_32", 2, 20, 24, 64, 0, 32, 3, 1, 1),
    ("rcu3x3_256", 1, 10, 10, 256, 0, 256, 3, 1, 1),
    ("proc3x3_768", 1, 10, 12, 768, 0, 256, 3, 1, 1),
    ("cat3x3_256+64", 1, 16, 16, 256, 64, 64, 3, 1, 1),
    ("patch7x7s4_c4", 2, 64, 64, 4, 0, 64, 7, 4, 3),
    ("ll7x7s2_c4", 1, 40, 40, 4, 0, 64, 7, 2, 3),
    ("patch3x3s2", 1, 20, 20, 64, 0, 128, 3, 2, 1),
    ("sr8x8s8", 1, 80, 80, 64, 0, 64, 8, 8, 0),
    ("sr2x2s2_320", 1, 20, 20, 320, 0, 320, 2, 2, 0),
    ("stem4x4s4_c4", 1, 32, 32, 4, 0, 96, 4, 4, 0),
    ("ds2x2s2_96", 1, 20, 20, 96, 0, 192, 2, 2, 0),
    ("odd_sizes", 3, 7, 5, 32, 0, 40, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_all_tiles(ops, case):
    name, B, H, W, C1, C2, Cout, K, stride, pad = case
    x = _rand((B, H, W, C1), 1)
    x2 = _rand((B, H, W, C2), 2) if C2 else None
    if C1 == 4:
        x[..., 3] = 0  # padded channel of the 3-channel inputs
    w = _rand((Cout, C1 + C2, K, K), 3, 1.0 / math.sqrt((C1 + C2) * K * K))
    b = _rand((Cout,), 4, 0.1)
    ref = _ref_conv(x, w, b, stride, pad, x2)
    xd = x.cuda()
    x2d = x2.cuda() if x2 is not None else None
    # both parity schemes of the split tiles: 0 = split-f16 (default), 3 = exact bf16 split (the fp32 tiles ignore it)
    for precision in (0, 3):
        for tile in [-1] + list(range(len(ops.conv_tiles()))):
            got = ops.conv2d(xd, w, b, stride=stride, pad=pad, x2=x2d, tile=tile, precision=precision)
            _close(got, ref, 2e-5, f"{name} tile {tile} precision {precision}")


def test_split_planes_are_lossless(ops):
    """fp32 -> three bf16 planes -> fp32 is the identity (8 + 8 + 8 significant bits), including tiny / huge magnitudes."""
    x = _rand((3, 17, 5, 64), 40)
    x.view(-1)[:64] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.0e-30, 1.17549435e-38] * 8)
    x.view(-1)[64:128] *= 1e-20
    xd = x.cuda()
    back = ops.split_planes(xd).merge()
    assert torch.equal(back, xd)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[4] % 32 == 0 and c[5] % 32 == 0], ids=[c[0] for c in CONV_CASES if c[4] % 32 == 0 and c[5] % 32 == 0])
def test_conv2d_split_plane_operands(ops, case):
    """The split-bf16 kernel fed with pre-split planes (and writing planes) must reproduce ITS OWN fp32-operand result
    bit for bit on every split tile: the planes are a lossless re-encoding and the MFMA sequence is the same."""
    name, B, H, W, C1, C2, Cout, K, stride, pad = case
    x = _rand((B, H, W, C1), 1).cuda()
    x2 = _rand((B, H, W, C2), 2).cuda() if C2 else None
    w = _rand((Cout, C1 + C2, K, K), 3, 1.0 / math.sqrt((C1 + C2) * K * K))
    b = _rand((Cout,), 4, 0.1)
    ref = _ref_conv(x.cpu(), w, b, stride, pad, None if x2 is None else x2.cpu())
    names = ops.conv_tiles()
    sb = [i for i, n in enumerate(names) if n.startswith("sb") and not n.startswith("sbh")]  # the halo tiles take fp32 operands only
    assert sb
    for tile in sb:
        # (splitk=False: split-K, which only fp32-in / fp32-out launches use, changes the summation order)
        base = ops.conv2d(x, w, b, stride=stride, pad=pad, x2=x2, tile=tile, act=1, precision=3, splitk=False)  # the exact bf16 split (planes are its format)
        got_in = ops.conv2d(x, w, b, stride=stride, pad=pad, x2=x2, tile=tile, act=1, planes_in=True, splitk=False)
        assert torch.equal(got_in, base), f"{name} {names[tile]}: split-plane input differs from fp32 input"
        if Cout % 4 == 0:
            got_io = ops.conv2d(x, w, b, stride=stride, pad=pad, x2=x2, tile=tile, act=1, planes_in=True, planes_out=True, splitk=False)
            assert torch.equal(got_io, base), f"{name} {names[tile]}: split-plane output differs"
    # automatic tile choice with plane operands only (no fp32 tile can run) + oracle check
    got = ops.conv2d(x, w, b, stride=stride, pad=pad, x2=x2, planes_in=True, planes_out=Cout % 4 == 0)
    _close(got, ref, 2e-5, f"{name} planes auto")
    # an exact-fp32 tile must refuse plane-only input loudly
    from perspectivefields_amd.engine import PfError
    with pytest.raises(PfError):
        ops.conv2d(x, w, b, stride=stride, pad=pad, x2=x2, tile=2, planes_in=True)


def test_conv2d_split_f16_scheme(ops):
    """The default parity scheme of the split tiles (2-way fp16 split, 3 MFMAs per product; include/pf_hip.h
    PF_PRECISION_FP32) on every split tile incl. the halo tiles, against fp64:
      * wide dynamic range (|x| over 2^+-12, per-channel weight magnitudes over 2^+-20: the per-channel power-of-two
        weight scale) -- error within 4x of the exact bf16 split (PF_PRECISION_FP32_BF16X6) + an fp32-rounding floor;
      * small activations: below |x| = 2^-3 the low part is an fp16 subnormal and the error per element turns ABSOLUTE, 2^-25 (sb_split.h) -- a tensor of
        1e-3 values keeps ~2^-15 relative accuracy, 1e-6 values ~2^-5: tensors that small need precision "fp32_bf16x6" (pf_check_range reports them);
      * |x| beyond the fp16 range saturates to +-65504 (finite output), everything below 65504 is exact-ish."""
    B, H, W, C, Cout = 2, 12, 16, 256, 256
    g = torch.Generator().manual_seed(123)
    x = _rand((B, H, W, C), 80) * torch.exp2(torch.randint(-12, 13, (B, H, W, C), generator=g).float())
    w = _rand((Cout, C, 3, 3), 81, 1.0 / math.sqrt(C * 9)) * torch.exp2(torch.randint(-20, 21, (Cout, 1, 1, 1), generator=g).float())
    b = _rand((Cout,), 82, 0.1)
    ref = _ref_conv(x, w, b, 1, 1)
    scale = _ref_conv(x.abs(), w.abs(), None, 1, 1)  # sum |x||w|: the natural error scale of a dot product
    xd = x.cuda()
    names = ops.conv_tiles()
    sb = [i for i, n in enumerate(names) if n.startswith("sb")]
    assert any(names[i].startswith("sbh") for i in sb)
    worst = 0.0
    for tile in sb:
        e16 = ((ops.conv2d(xd, w, b, pad=1, tile=tile, precision=0).double().cpu() - ref).abs() / scale).max().item()
        e6 = ((ops.conv2d(xd, w, b, pad=1, tile=tile, precision=3).double().cpu() - ref).abs() / scale).max().item()
        worst = max(worst, e16)
        assert e16 <= 4 * e6 + 2.0 ** -22, (names[tile], e16, e6)
        assert e16 <= 2.0 ** -17, (names[tile], e16)  # (this data's fp32 accumulation error alone -- e6 -- reaches ~1e-6)
    print(f"[split-f16] worst |err| / sum|x||w| over {len(sb)} tiles: {worst:.2e} (2^-22 = {2.0 ** -22:.2e})")
    # small / tiny activations (explicit split tile: tile -1 may pick any family)
    t64 = names.index("sb64x64")
    wt = _rand((64, 64, 1, 1), 84, 0.125)
    for a_scale in (1.0, 1e-3, 1e-6):
        xt = (_rand((1, 8, 8, 64), 83) * a_scale).cuda()
        rt = _ref_conv(xt.cpu(), wt, None, 1, 0)
        got = ops.conv2d(xt, wt, None, precision=0, tile=t64).double().cpu()
        bound = 3 * 2.0 ** -22 * _ref_conv(xt.cpu().abs(), wt.abs(), None, 1, 0) + 2.0 ** -25 * wt.abs().double().sum(dim=(1, 2, 3)).view(1, 1, 1, -1)
        assert bool(((got - rt).abs() <= bound).all()), a_scale
        # the exact bf16 split has no such window
        e6 = ops.conv2d(xt, wt, None, precision=3, tile=t64).double().cpu()
        assert ((e6 - rt).abs() / _ref_conv(xt.cpu().abs(), wt.abs(), None, 1, 0)).max().item() <= 2.0 ** -20, a_scale
    # saturation at the fp16 range
    xs = torch.zeros((1, 4, 4, 32)); xs[..., 0] = 1.0e6; xs[..., 1] = -70000.0; xs[..., 2] = 65000.0
    ws = torch.zeros((32, 32, 1, 1)); ws[0, 0] = ws[1, 1] = ws[2, 2] = 1.0
    ys = ops.conv2d(xs.cuda(), ws, None, precision=0, tile=t64).cpu()
    assert torch.isfinite(ys).all()
    assert torch.allclose(ys[..., 0], torch.tensor(65504.0)) and torch.allclose(ys[..., 1], torch.tensor(-65504.0)) and torch.allclose(ys[..., 2], torch.tensor(65000.0))
    # ... while the exact bf16 split has no range restriction
    assert torch.allclose(ops.conv2d(xs.cuda(), ws, None, precision=3, tile=t64).cpu()[..., 0], torch.tensor(1.0e6))


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[4] % 32 == 0 and c[5] % 32 == 0], ids=[c[0] for c in CONV_CASES if c[4] % 32 == 0 and c[5] % 32 == 0])
def test_conv2d_split_f16_plane_operands(ops, case):
    """The split-f16 scheme fed with the two fp16 planes a producer kernel wrote (format bit 1, sb_split.h) instead of fp32:
    the planes hold exactly what the GEMM's own staging would compute, so the results are bit-identical on every linear
    split tile; plane OUTPUT is the fp32 result re-split (hi + lo: within 2^-22 of it, 2^-25 absolute below 2^-3)."""
    name, B, H, W, C1, C2, Cout, K, stride, pad = case
    x = _rand((B, H, W, C1), 1).cuda()
    x2 = _rand((B, H, W, C2), 2).cuda() if C2 else None
    w = _rand((Cout, C1 + C2, K, K), 3, 1.0 / math.sqrt((C1 + C2) * K * K))
    b = _rand((Cout,), 4, 0.1)
    names = ops.conv_tiles()
    halo_ok = K == 3 and stride == 1 and pad == 1 and C2 == 0  # the halo kernel copies fp16 planes too (igemm_sbh ASB): one input, plain tap loop
    for tile in [i for i, n in enumerate(names) if n.startswith("sb") and (halo_ok or not n.startswith("sbh"))]:
        base = ops.conv2d(x, w, b, stride=stride, pad=pad, x2=x2, tile=tile, act=1, precision=0, splitk=False)
        got_in = ops.conv2d(x, w, b, stride=stride, pad=pad, x2=x2, tile=tile, act=1, planes_in=True, planes_fmt="f16x2", splitk=False)
        assert torch.equal(got_in, base), f"{name} {names[tile]}: fp16-plane input differs from fp32 input"
        if Cout % 4 == 0:
            got_io = ops.conv2d(x, w, b, stride=stride, pad=pad, x2=x2, tile=tile, act=1, planes_in=True, planes_out=True, planes_fmt="f16x2", splitk=False)
            assert bool(((got_io - base).abs() <= 2.0 ** -22 * base.abs() + 2.0 ** -25).all()), f"{name} {names[tile]}: fp16-plane output"
    # round trip of the format itself: 22+ significant bits from 2^-3 up to the fp16 range, 2^-25 absolute below, saturation outside
    v = _rand((4, 8, 8, 32), 5) * torch.exp2(torch.randint(-10, 12, (4, 8, 8, 32)).float())
    back = ops.split_planes(v.cuda(), "f16x2").merge().cpu()
    assert bool(((back - v).abs() <= 2.0 ** -22 * v.abs() + 2.0 ** -25).all())


def test_split_planes_bits(ops):
    """The device split of the split-f16 scheme (sb_split.h split2_f16: v_med3 + v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16 with an fp16 source operand) against its
    definition computed on the CPU, BIT for BIT: hi = fp16_rn(clamp(x)), lo = fp16_rn(clamp(x) - hi), subnormal low parts kept, saturation at +-65504."""
    g = torch.Generator().manual_seed(11)
    v = _rand((64, 33, 32), 6) * torch.exp2(torch.randint(-30, 17, (64, 33, 32), generator=g).float())
    v.view(-1)[:16] = torch.tensor([0.0, -0.0, 65504.0, -65504.0, 65520.0, 1e9, -1e9, 6.1e-5, 5.9e-8, 1e-10, 0.125, 0.1249999, 2048.5, 1.0009766, -3.0e38, 7.0e4])
    n = v.numel()
    pl = ops.split_planes(v.cuda(), "f16x2")
    raw = pl.data.cpu()
    stride = pl.plane_elems & ~1
    hi_dev, lo_dev = raw[:n], raw[stride:stride + n]
    c = v.reshape(-1).clamp(-65504.0, 65504.0)
    hi = c.to(torch.float16)
    lo = (c - hi.float()).to(torch.float16)
    assert torch.equal(hi_dev, hi.view(torch.int16)), "hi plane differs from fp16_rn(clamp(x))"
    bad = (lo_dev != lo.view(torch.int16)) & ~((lo == 0) & (lo_dev.view(torch.float16) == 0))  # +0 / -0 of an exact remainder may differ in sign
    assert not bool(bad.any()), f"lo plane differs in {int(bad.sum())} of {n} elements, first at {int(bad.nonzero()[0])}: x = {float(v.reshape(-1)[bad.nonzero()[0]])}"


def test_elementwise_plane_outputs(ops):
    """LayerNorm / depthwise+GELU / attention / bilinear x2 writing split planes == their fp32 output, exactly."""
    x = _rand((2, 20, 20, 320), 50).cuda()
    g, be = _rand((320,), 51), _rand((320,), 52, 0.1)
    assert torch.equal(ops.layernorm(x, g, be, 1e-6, planes_out=True), ops.layernorm(x, g, be, 1e-6))
    xh = _rand((2, 20, 20, 1280), 53).cuda()
    wd, bd = _rand((1280, 1, 3, 3), 54, 0.3), _rand((1280,), 55, 0.1)
    assert torch.equal(ops.dwconv3x3_gelu(xh, wd, bd, planes_out=True), ops.dwconv3x3_gelu(xh, wd, bd))
    for (B, H, W, C) in [(1, 80, 80, 256), (2, 9, 13, 128)]:
        xs = _rand((B, H, W, C), 56).cuda()
        wd, bd = _rand((C, 1, 3, 3), 57, 0.3), _rand((C,), 58, 0.1)
        assert torch.equal(ops.dwconv3x3_gelu(xs, wd, bd, planes_out=True), ops.dwconv3x3_gelu(xs, wd, bd))
    q, kv = _rand((2, 400, 320), 59).cuda(), _rand((2, 100, 640), 60).cuda()
    assert torch.equal(ops.sr_attention(q, kv, 5, planes_out=True), ops.sr_attention(q, kv, 5))
    xu = _rand((2, 10, 12, 64), 61).cuda()
    assert torch.equal(ops.upsample2x(xu, planes_out=True), ops.upsample2x(xu))


def test_conv2d_halo_tiles_partial_patches(ops):
    """The 3x3 halo-tile kernels own 8 x 16 output patches: maps that are not multiples of the patch, one-pixel maps,
    batch > 1, ragged Cout, concat input, residual / bias-table-free epilogue -- against the fp64 reference."""
    names = ops.conv_tiles()
    halo = [i for i, n in enumerate(names) if n.startswith("sbh")]
    assert halo
    for (B, H, W, C1, C2, Cout) in [(2, 10, 10, 64, 0, 256), (1, 23, 37, 32, 0, 40), (3, 8, 16, 96, 32, 64), (1, 1, 1, 32, 0, 32), (2, 40, 24, 128, 0, 128)]:
        x = _rand((B, H, W, C1), 81)
        x2 = _rand((B, H, W, C2), 82) if C2 else None
        w = _rand((Cout, C1 + C2, 3, 3), 83, 1.0 / math.sqrt((C1 + C2) * 9))
        b = _rand((Cout,), 84, 0.1)
        r1 = _rand((B, H, W, Cout), 85)
        ref = F.relu(_ref_conv(x, w, b, 1, 1, x2) + r1.double())
        for tile in halo:
            got = ops.conv2d(x.cuda(), w, b, pad=1, x2=None if x2 is None else x2.cuda(), res1=r1.cuda(), post_relu=True, tile=tile)
            _close(got, ref, 2e-5, f"{names[tile]} {B}x{H}x{W} {C1}+{C2}->{Cout}")


def test_conv2d_epilogues(ops):
    B, H, W, C = 2, 12, 12, 256
    x = _rand((B, H, W, C), 5)
    w = _rand((C, C, 3, 3), 6, 1.0 / math.sqrt(C * 9))
    b = _rand((C,), 7, 0.1)
    r1 = _rand((B, H, W, C), 8)
    r2 = _rand((B, H, W, C), 9)
    base = _ref_conv(x, w, b, 1, 1)
    xd, r1d, r2d = x.cuda(), r1.cuda(), r2.cuda()
    _close(ops.conv2d(xd, w, b, pad=1, act=1), F.relu(base), 2e-5, "relu")
    _close(ops.conv2d(xd, w, b, pad=1, act=2), pf_oracle.gelu(base), 2e-5, "gelu")
    _close(ops.conv2d(xd, w, b, pad=1, res1=r1d), base + r1.double(), 2e-5, "res1")
    _close(ops.conv2d(xd, w, b, pad=1, res1=r1d, res2=r2d, post_relu=True), F.relu(base + r1.double() + r2.double()), 2e-5, "res1+res2+relu")
    _close(ops.conv2d(xd, w, None, pad=1), base - b.double(), 2e-5, "no bias")
    # in-place residual (y aliases res1), as the engine uses for the transformer stream
    y = r1.cuda().clone()
    from perspectivefields_amd.engine import load_library, _stream_ptr
    got = ops.conv2d(xd, w, b, pad=1, res1=y)
    _close(got, base + r1.double(), 2e-5, "res1 (separate out)")


def test_conv2d_nchw_logits(ops):
    """1x1 conv to 73 / 180 classes with the NCHW store used for the API-visible logits."""
    B, H, W = 2, 16, 24
    x = _rand((B, H, W, 32), 10)
    for n in (73, 180):
        w = _rand((n, 32, 1, 1), 11, 0.2)
        b = _rand((n,), 12, 0.1)
        ref = _ref_conv(x, w, b, 1, 0).permute(0, 3, 1, 2).contiguous()
        for tile in [-1] + list(range(len(ops.conv_tiles()))):
            got = ops.conv2d(x.cuda(), w, b, nchw_out=True, tile=tile)
            _close(got, ref, 2e-5, f"nchw n={n} tile {tile}")


@pytest.mark.parametrize("K,N,rows", [(64, 256, 300), (320, 1280, 257), (1280, 320, 129), (2048, 512, 100), (96, 384, 64), (3072, 768, 33), (32, 2, 200)])
def test_linear(ops, K, N, rows):
    x = _rand((rows, K), 13)
    w = _rand((N, K), 14, 1.0 / math.sqrt(K))
    b = _rand((N,), 15, 0.1)
    r = _rand((rows, N), 16)
    ref = F.linear(x.double(), w.double(), b.double())
    _close(ops.linear(x.cuda(), w, b), ref, 3e-5, "linear")
    _close(ops.linear(x.cuda(), w, b, act=2), pf_oracle.gelu(ref), 3e-5, "linear+gelu")
    _close(ops.linear(x.cuda(), w, b, res1=r.cuda()), ref + r.double(), 3e-5, "linear+res")


@pytest.mark.parametrize("K,N,rows", [(64, 256, 300), (320, 1280, 257), (320, 640, 100), (512, 1024, 129), (96, 384, 200), (768, 3072, 33), (128, 512, 1000)])
def test_linear_with_fused_layernorm(ops, K, N, rows):
    """Linear(LayerNorm(x)) with the LayerNorm inside the GEMM (ConvParams::ln): every linear split tile and every contraction scheme,
    rows with a large common offset (the per-row pivot must keep the one-pass variance and the mean correction accurate), GELU and
    residual epilogues.  Oracle: torch fp64."""
    x = _rand((rows, K), 31, 1.5)
    x = x + 40.0 * _rand((rows, 1), 32)          # per-row offsets up to 40 sigma
    x[3] = 0.0                                    # a constant row: variance 0 -> rstd = eps^-1/2, output = bias term
    w = _rand((N, K), 33, 1.0 / math.sqrt(K))
    b = _rand((N,), 34, 0.1)
    g = 1 + _rand((K,), 35, 0.3)
    be = _rand((K,), 36, 0.2)
    r = _rand((rows, N), 37)
    for eps in (1e-6, 1e-5):
        xn = F.layer_norm(x.double(), (K,), g.double(), be.double(), eps)
        ref = F.linear(xn, w.double(), b.double())
        xd = x.cuda()
        _close(ops.linear_ln(xd, w, b, g, be, eps), ref, 5e-5, "linear_ln")
        _close(ops.linear_ln(xd, w, b, g, be, eps, act=2), pf_oracle.gelu(ref), 5e-5, "linear_ln+gelu")
        _close(ops.linear_ln(xd, w, b, g, be, eps, res1=r.cuda()), ref + r.double(), 5e-5, "linear_ln+res")
    tiles = ops.conv_tiles()
    ref = F.linear(F.layer_norm(x.double(), (K,), g.double(), be.double(), 1e-6), w.double(), b.double())
    ran = 0
    for t, name in enumerate(tiles):
        if not name.startswith("sb") or name.startswith("sbh"):
            with pytest.raises(Exception):
                ops.linear_ln(x.cuda(), w, b, g, be, 1e-6, tile=t)   # exact-fp32 and halo tiles do not carry the fused form: loud
            continue
        for prec, tol in ((0, 5e-5), (3, 5e-5)):
            _close(ops.linear_ln(x.cuda(), w, b, g, be, 1e-6, tile=t, precision=prec), ref, tol, f"linear_ln tile {name} precision {prec}")
            ran += 1
    assert ran >= 20


def _with_outlier_channels(x, seed, sigma=100.0):
    """Rows as trained transformers produce them (VERDICT r02 / ADVICE): a third of the rows get a `sigma`-sized outlier in channel 0, a third in a random
    channel, a third both with opposite signs -- a pivot taken from one channel would turn that into a common offset of the whole shifted row."""
    x = x.clone()
    flat = x.reshape(-1, x.shape[-1])
    g = torch.Generator().manual_seed(seed)
    ch = torch.randint(0, flat.shape[1], (flat.shape[0],), generator=g)
    sgn = torch.where(torch.rand(flat.shape[0], generator=g) < 0.5, -1.0, 1.0) * sigma * float(flat.std())
    r = torch.arange(flat.shape[0])
    a, b = r % 3 == 0, r % 3 == 1
    flat[a | ~(a | b), 0] += sgn[a | ~(a | b)]
    sel = b | ~(a | b)
    flat[r[sel], ch[sel]] -= sgn[sel]
    return x


@pytest.mark.parametrize("K,N,rows", [(64, 256, 300), (320, 1280, 257), (512, 1024, 129), (96, 384, 200), (768, 3072, 33)])
def test_linear_with_fused_layernorm_outlier_channels(ops, K, N, rows):
    """The fused LayerNorm with 100-sigma outlier channels (in channel 0, in a random channel, in both): the per-row pivot is the mean of the first staged
    32-channel chunk, not one channel, so neither the one-pass variance nor `acc - mean * colsum` cancels.  Every linear split tile; oracle torch fp64."""
    x = _with_outlier_channels(_rand((rows, K), 131, 1.5) + 10.0 * _rand((rows, 1), 132), 133)
    w = _rand((N, K), 134, 1.0 / math.sqrt(K))
    b = _rand((N,), 135, 0.1)
    g = 1 + _rand((K,), 136, 0.3)
    be = _rand((K,), 137, 0.2)
    ref = F.linear(F.layer_norm(x.double(), (K,), g.double(), be.double(), 1e-6), w.double(), b.double())
    tiles = ops.conv_tiles()
    ran = 0
    for t, name in enumerate(tiles):
        if not name.startswith("sb") or name.startswith("sbh") or name.startswith(("sbA", "sbI_", "sbPI_")):
            continue
        _close(ops.linear_ln(x.cuda(), w, b, g, be, 1e-6, tile=t), ref, 5e-5, f"linear_ln with outlier channels, tile {name}")
        ran += 1
    assert ran >= 8


@pytest.mark.parametrize("C,rows", [(96, 1000), (192, 300)])
def test_convnext_block_mlp_fused_outlier_channels(ops, C, rows):
    d = _with_outlier_channels(_rand((rows, C), 141, 1.5) + 10.0 * _rand((rows, 1), 142), 143)
    y = _rand((rows, C), 43)
    w1, b1 = _rand((4 * C, C), 44, 1.0 / math.sqrt(C)), _rand((4 * C,), 45, 0.1)
    g, be = 1 + _rand((C,), 46, 0.3), _rand((C,), 47, 0.2)
    w2, b2, ls = _rand((C, 4 * C), 48, 1.0 / math.sqrt(4 * C)), _rand((C,), 49, 0.1), _rand((C,), 50, 0.5)
    h = pf_oracle.gelu(F.linear(F.layer_norm(d.double(), (C,), g.double(), be.double(), 1e-6), w1.double(), b1.double()))
    ref = y.double() + ls.double() * F.linear(h, w2.double(), b2.double())
    _close(ops.cnx_mlp(d.cuda(), y.cuda(), w1, b1, g, be, 1e-6, w2, b2, ls), ref, 5e-5, f"fused ConvNeXt MLP with outlier channels C={C}")


@pytest.mark.parametrize("C,B,Hs,Ws", [(64, 1, 24, 40), (128, 1, 16, 16)])
def test_mit_block_mlp_fused_outlier_channels(ops, C, B, Hs, Ws):
    x = _with_outlier_channels(_rand((B, Hs, Ws, C), 151, 1.5) + 10.0 * _rand((B, Hs, Ws, 1), 152), 153)
    w1, b1 = _rand((4 * C, C), 53, 1.0 / math.sqrt(C)), _rand((4 * C,), 54, 0.1)
    g, be = 1 + _rand((C,), 55, 0.3), _rand((C,), 56, 0.2)
    wd, bd = _rand((4 * C, 1, 3, 3), 57, 0.4), _rand((4 * C,), 58, 0.1)
    w2, b2 = _rand((C, 4 * C), 59, 1.0 / math.sqrt(4 * C)), _rand((C,), 60, 0.1)
    h = F.linear(F.layer_norm(x.double(), (C,), g.double(), be.double(), 1e-6), w1.double(), b1.double())
    h = F.conv2d(h.permute(0, 3, 1, 2), wd.double(), bd.double(), padding=1, groups=4 * C).permute(0, 2, 3, 1)
    ref = x.double() + F.linear(pf_oracle.gelu(h), w2.double(), b2.double())
    _close(ops.mit_mlp(x.cuda(), w1, b1, g, be, 1e-6, wd, bd, w2, b2), ref, 5e-5, f"fused MiT Mlp with outlier channels C={C}")


@pytest.mark.parametrize("C,rows", [(96, 128), (96, 1000), (96, 6400), (192, 300), (192, 1600)])
def test_convnext_block_mlp_fused(ops, C, rows):
    """cnx_mlp.hip: y + ls * pwconv2(GELU(pwconv1(LayerNorm(d)))) in one kernel (hidden map in registers) vs torch fp64 (convnext.py:49-58).
    Rows with large common offsets, a constant row, ragged row count (tail block)."""
    d = _rand((rows, C), 41, 1.5) + 30.0 * _rand((rows, 1), 42)
    d[5] = 2.0
    y = _rand((rows, C), 43)
    w1 = _rand((4 * C, C), 44, 1.0 / math.sqrt(C))
    b1 = _rand((4 * C,), 45, 0.1)
    g = 1 + _rand((C,), 46, 0.3)
    be = _rand((C,), 47, 0.2)
    w2 = _rand((C, 4 * C), 48, 1.0 / math.sqrt(4 * C))
    b2 = _rand((C,), 49, 0.1)
    ls = _rand((C,), 50, 0.5)
    xn = F.layer_norm(d.double(), (C,), g.double(), be.double(), 1e-6)
    h = pf_oracle.gelu(F.linear(xn, w1.double(), b1.double()))
    ref = y.double() + ls.double() * F.linear(h, w2.double(), b2.double())
    got = ops.cnx_mlp(d.cuda(), y.cuda(), w1, b1, g, be, 1e-6, w2, b2, ls)
    _close(got, ref, 5e-5, f"fused ConvNeXt MLP C={C}")
    # and against the two-GEMM form it replaces (LayerNorm-fused pwconv1 + pwconv2 with the layer scale folded): same scheme, ~1e-6
    hid = ops.linear_ln(d.cuda(), w1, b1, g, be, 1e-6, act=2)
    two = ops.linear(hid, w2 * ls[:, None], b2 * ls, res1=y.cuda())
    _close(got, two.double().cpu(), 2e-5, f"fused ConvNeXt MLP vs two GEMMs C={C}")


@pytest.mark.parametrize("C,B,Hs,Ws", [(64, 2, 80, 80), (64, 1, 13, 21), (64, 3, 8, 16), (128, 2, 40, 40), (128, 1, 11, 9), (128, 2, 8, 8)])
def test_mit_block_mlp_fused(ops, C, B, Hs, Ws):
    """mit_mlp.hip: x + fc2(GELU(dwconv3x3(fc1(LayerNorm(x))))) in one kernel (hidden map in LDS / registers) vs torch fp64
    (mix_transformers.py:49-56,200,497-508).  Full-size and ragged maps (patches cut by the border, maps smaller than a patch + halo)."""
    x = _rand((B, Hs, Ws, C), 51, 1.5) + 20.0 * _rand((B, Hs, Ws, 1), 52)
    w1 = _rand((4 * C, C), 53, 1.0 / math.sqrt(C))
    b1 = _rand((4 * C,), 54, 0.1)
    g = 1 + _rand((C,), 55, 0.3)
    be = _rand((C,), 56, 0.2)
    wd = _rand((4 * C, 1, 3, 3), 57, 0.4)
    bd = _rand((4 * C,), 58, 0.1)
    w2 = _rand((C, 4 * C), 59, 1.0 / math.sqrt(4 * C))
    b2 = _rand((C,), 60, 0.1)
    xn = F.layer_norm(x.double(), (C,), g.double(), be.double(), 1e-6)
    h = F.linear(xn, w1.double(), b1.double())                                   # (B, Hs, Ws, 4C)
    h = F.conv2d(h.permute(0, 3, 1, 2), wd.double(), bd.double(), padding=1, groups=4 * C).permute(0, 2, 3, 1)
    ref = x.double() + F.linear(pf_oracle.gelu(h), w2.double(), b2.double())
    got = ops.mit_mlp(x.cuda(), w1, b1, g, be, 1e-6, wd, bd, w2, b2)
    _close(got, ref, 5e-5, f"fused MiT Mlp C={C} {Hs}x{Ws}")


def test_mfma_operand_orientation(ops):
    """A = I-like check with an ASYMMETRIC B: catches a row/col swap of the MFMA C layout."""
    K = N = 64
    rows = 64
    x = torch.zeros(rows, K)
    x[torch.arange(rows), torch.arange(rows) % K] = 1.0
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 100.0
    ref = F.linear(x.double(), w.double())
    _close(ops.linear(x.cuda(), w), ref, 1e-6, "identity A, asymmetric B")


@pytest.mark.parametrize("C,eps", [(64, 1e-5), (96, 1e-6), (128, 1e-6), (192, 1e-6), (320, 1e-6), (384, 1e-6), (512, 1e-5), (768, 1e-6)])
def test_layernorm(ops, C, eps):
    x = _rand((37, 5, C), 17, 2.0) + 0.5
    g = 1 + _rand((C,), 18, 0.1)
    b = _rand((C,), 19, 0.1)
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), eps)
    _close(ops.layernorm(x.cuda(), g, b, eps), ref, 1e-5, f"layernorm C={C}")


@pytest.mark.parametrize("B,H,W,C", [(2, 80, 80, 256), (1, 40, 40, 512), (2, 20, 20, 1280), (3, 10, 10, 2048), (1, 9, 13, 128), (2, 11, 16, 128), (1, 21, 20, 256)])
def test_dwconv3x3_gelu(ops, B, H, W, C):
    x = _rand((B, H, W, C), 20)
    w = _rand((C, 1, 3, 3), 21, 0.4)
    b = _rand((C,), 22, 0.1)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1, groups=C)
    ref = pf_oracle.gelu(ref).permute(0, 2, 3, 1).contiguous()
    xd = x.cuda()
    y0 = ops.dwconv3x3_gelu(xd, w, b)
    _close(y0, ref, 1e-5, "dwconv3x3+gelu")
    # multi-column / prefetching kernel (elem.hip dwconv3x3_gelu_mc_kernel): every (block shape, strip height, columns x prefetch)
    # the shape fits -- same tap order, so bit-identical to the one-column kernel; also with split-plane output
    for shape, (cqb, xb) in enumerate(((32, 8), (64, 4), (64, 2), (64, 5))):
        for code, (nc, pf) in enumerate(((1, 1), (2, 0), (2, 1), (2, 2), (1, 2))):
            if (C // 4) % cqb or W % (xb * nc):
                continue
            for th in range(3):
                v = 1000 + 100 * shape + 10 * th + code
                assert torch.equal(ops.dwconv3x3_gelu(xd, w, b, variant=v), y0), f"dwconv3x3 mc variant {v}"
            assert torch.equal(ops.dwconv3x3_gelu(xd, w, b, variant=1000 + 100 * shape + code, planes_out=True), ops.dwconv3x3_gelu(xd, w, b, planes_out=True)), "dwconv3x3 mc planes"


@pytest.mark.parametrize("B,H,W,C", [(2, 80, 80, 96), (1, 40, 40, 192), (2, 20, 20, 384), (3, 10, 10, 768), (1, 2, 2, 768), (1, 5, 11, 96), (2, 20, 13, 96), (1, 33, 20, 96), (1, 17, 20, 96), (2, 15, 19, 96), (1, 14, 7, 96), (2, 40, 13, 96), (1, 20, 9, 96)])
def test_dwconv7x7(ops, B, H, W, C):
    x = _rand((B, H, W, C), 23)
    w = _rand((C, 1, 7, 7), 24, 0.15)
    b = _rand((C,), 25, 0.1)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=3, groups=C).permute(0, 2, 3, 1).contiguous()
    xd = x.cuda()
    _close(ops.dwconv7x7(xd, w, b), ref, 1e-5, "dwconv7x7")
    # every kernel / configuration: column-blocked (nc x nb x strip heights incl. one-strip and tiny strips), one column per lane
    for nc in (4, 2):
        for nb in (2, 3):
            for th in (0, 1, 3, 7, H):
                _close(ops.dwconv7x7(xd, w, b, variant=3, nc=nc, nb=nb, th=th), ref, 1e-5, f"dwconv7x7 cb nc{nc} nb{nb} th{th}")
    _close(ops.dwconv7x7(xd, w, b, variant=2), ref, 1e-5, "dwconv7x7 lane")
    # LDS-tile kernel (maps of <= 20 columns): strips of every kind, same accumulation order as the streaming kernel -> identical bits
    cb = ops.dwconv7x7(xd, w, b, variant=3)
    assert torch.equal(ops.dwconv7x7(xd, w, b), cb), "the default path (packed kernels) differs from the scalar streaming kernel"
    for th in (0, 1, 3, 7, H):
        got = ops.dwconv7x7(xd, w, b, variant=4, th=th)
        _close(got, ref, 1e-5, f"dwconv7x7 lds th{th}")
        assert torch.equal(got, cb), f"dwconv7x7 lds th{th} differs from the streaming kernel"
    # packed-fp32 forms (dw7_pk.hip): the same fused multiply-adds in the same order, two output columns per v_pk_fma_f32 -> identical bits.  Streaming kernel (run-time
    # strips and the straight-line strips of 10 / 20 rows) and the tile-in-parts LDS kernel (32 / 16 channels per block; configurations a shape does not admit fall
    # back inside the library and must still be right)
    for nc in (4, 2):
        for nb in (2, 3):
            for th in (0, 1, 3, 7, 10, 20, H):
                got = ops.dwconv7x7(xd, w, b, variant=5, nc=nc, nb=nb, th=th)
                assert torch.equal(got, cb), f"dwconv7x7 packed cb nc{nc} nb{nb} th{th} differs from the scalar kernel"
    for ch in (0, 32, 16):
        for th in (0, 5, 10, 20):
            got = ops.dwconv7x7(xd, w, b, variant=6, nc=ch, th=th)
            assert torch.equal(got, cb), f"dwconv7x7 packed lds ch{ch} th{th} differs from the scalar kernel"


@pytest.mark.parametrize("B,N,heads,M", [(2, 6400, 1, 100), (1, 1600, 2, 100), (2, 400, 5, 100), (3, 100, 8, 100), (1, 70, 2, 37)])
def test_sr_attention(ops, B, N, heads, M):
    C = heads * 64
    q = _rand((B, N, C), 26)
    kv = _rand((B, M, 2 * C), 27)
    qh = q.double().reshape(B, N, heads, 64).transpose(1, 2)
    k = kv.double()[..., :C].reshape(B, M, heads, 64).transpose(1, 2)
    v = kv.double()[..., C:].reshape(B, M, heads, 64).transpose(1, 2)
    a = ((qh @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    ref = (a @ v).transpose(1, 2).reshape(B, N, C)
    _close(ops.sr_attention(q.cuda(), kv.cuda(), heads), ref, 2e-5, "sr attention")


def _attention_bound(q, kv, heads):
    """fp64 reference and a per-row error bound for the split-f16 kernel: a logit carries <= 3 * 2^-22 * sum_d |q_d k_d| / 8
    (the split products) -> relative change of the softmax weights <= 2 delta; plus 2^-21 from the second product."""
    B, N, C = q.shape
    M = kv.shape[1]
    qh = q.double().reshape(B, N, heads, 64).transpose(1, 2)
    k = kv.double()[..., :C].reshape(B, M, heads, 64).transpose(1, 2)
    v = kv.double()[..., C:].reshape(B, M, heads, 64).transpose(1, 2)
    a = ((qh @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    ref = (a @ v).transpose(1, 2).reshape(B, N, C)
    # + the absolute floor of the scaled split (entries of K below 2^-6: 2^-29 each)
    delta = 3 * 2.0 ** -22 * 0.125 * (qh.abs() @ k.abs().transpose(-2, -1)).max(-1).values + 2.0 ** -29 * 0.125 * qh.abs().sum(-1)  # (B, heads, N)
    vmax = v.abs().amax(dim=(2, 3))  # (B, heads)
    bound = (2 * delta + 2.0 ** -20) * vmax[:, :, None] + 1e-7
    return ref, bound.transpose(1, 2)[..., None].expand(B, N, heads, 64).reshape(B, N, C)


@pytest.mark.parametrize("B,N,heads,M", [(2, 6400, 1, 100), (1, 1600, 2, 100), (2, 400, 5, 100), (3, 100, 8, 100), (1, 70, 2, 37), (1, 130, 1, 128), (1, 33, 1, 16)])
def test_sr_attention_both_kernels(ops, B, N, heads, M):
    """the split-f16 MFMA kernel (default) within its analytic bound, the exact-fp32 MFMA kernel at 1e-5, and the two agree"""
    C = heads * 64
    q = _rand((B, N, C), 26)
    kv = _rand((B, M, 2 * C), 27)
    ref, bound = _attention_bound(q, kv, heads)
    f16 = ops.sr_attention_variant(q.cuda(), kv.cuda(), heads, 1).double().cpu()
    f32 = ops.sr_attention_variant(q.cuda(), kv.cuda(), heads, 0).double().cpu()
    assert torch.isfinite(f16).all()
    worst = float(((f16 - ref).abs() / bound).max())
    print(f"[attention split-f16 B{B} N{N} h{heads} M{M}] max err {float((f16 - ref).abs().max()):.2e} (ratio to bound {worst:.2f}); fp32 kernel {float((f32 - ref).abs().max()):.2e}")
    assert worst <= 1.0
    _close(f32, ref, 1e-5, "attention fp32 MFMA")
    _close(f16, ref, 2e-5, "attention split-f16 (O(1) data)")


@pytest.mark.parametrize("B,N,M", [(2, 6400, 100), (3, 100, 100), (1, 70, 37), (1, 33, 16), (2, 300, 1), (1, 1000, 128)])
def test_mit_attn64_block(ops, B, N, M):
    """r06 (attn_block.hip): the attention half of a one-head MiT block -- LayerNorm-1, q projection, softmax(q k^T / 8) v, output projection, residual
    (mix_transformers.py:199 with :108-141) -- in ONE kernel, every intermediate in registers (transposed products with a permuted contraction index).  Oracle: torch fp64;
    rows with offsets and an outlier channel (LayerNorm), a dominating key (softmax max path), ragged tiles, in-place update."""
    C = 64
    x = _rand((B, N, C), 40) * 2.0 + 0.5
    x[0, 3] += 40.0
    x[0, 5, 7] = 300.0
    kv = _rand((B, M, 2 * C), 41)
    g, be = 1.0 + 0.2 * _rand((C,), 42), 0.1 * _rand((C,), 43)
    qw, qb = _rand((C, C), 44, 1.0 / 8.0), 0.1 * _rand((C,), 45)
    pw, pb = _rand((C, C), 46, 1.0 / 8.0), 0.1 * _rand((C,), 47)
    xd = x.double()
    xn = F.layer_norm(xd, (C,), g.double(), be.double(), 1e-6)
    q = xn @ qw.double().t() + qb.double()
    if M >= 100:
        kv[0, 17, :C] = q[0, 9].float() * 3.0     # one key dominates a row
    k, v = kv.double()[..., :C], kv.double()[..., C:]
    a = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    ref = xd + (a @ v) @ pw.double().t() + pb.double()
    got = ops.mit_attn64(x.cuda(), kv.cuda(), g, be, 1e-6, qw, qb, pw, pb)
    err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    _, ms = ops.mit_attn64(x.cuda(), kv.cuda(), g, be, 1e-6, qw, qb, pw, pb, iters=10)
    print(f"[mit_attn64 B{B} N{N} M{M}] max |err| / max |ref| {err:.2e}; {1e3 * ms:.1f} us per launch")
    _close(got, ref, 3e-5, "mit_attn64")
    xin = x.cuda().clone()
    assert torch.equal(ops.mit_attn64(xin, kv.cuda(), g, be, 1e-6, qw, qb, pw, pb, inplace=True), got)   # in place: a block reads and writes its own rows only


@pytest.mark.parametrize("B,H,W,stride,ln", [(2, 320, 320, 2, False), (2, 320, 320, 4, True), (1, 37, 53, 2, False), (3, 20, 9, 4, True), (1, 7, 7, 4, True)])
def test_stem7x7(ops, B, H, W, stride, ln):
    """r06 (stem7.hip): the two 7 x 7 convs that read the normalised image -- the low-level encoder (stride 2, BatchNorm folded, ReLU: perspectivefields.py:70-83) and the
    first patch embedding (stride 4 + LayerNorm eps 1e-5: mix_transformers.py:205-246) -- as one specialised kernel.  Oracle: torch fp64; image-range inputs (|x| <= 130),
    odd sizes (borders, ragged last tile)."""
    x = (_rand((B, H, W, 3), 60) * 60.0)
    x4 = torch.cat([x, torch.zeros(B, H, W, 1)], dim=-1)
    w = _rand((64, 3, 7, 7), 61, 1.0 / math.sqrt(147))
    b = 0.1 * _rand((64,), 62)
    g, be = (1.0 + 0.2 * _rand((64,), 63), 0.1 * _rand((64,), 64)) if ln else (None, None)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=3).permute(0, 2, 3, 1)
    ref = F.layer_norm(ref, (64,), g.double(), be.double(), 1e-5) if ln else torch.relu(ref)
    got = ops.stem7x7(x4.cuda(), w, b, stride, relu=not ln, ln_gamma=g, ln_beta=be, eps=1e-5)
    assert tuple(got.shape) == tuple(ref.shape)
    err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    _, ms = ops.stem7x7(x4.cuda(), w, b, stride, relu=not ln, ln_gamma=g, ln_beta=be, eps=1e-5, iters=10)
    print(f"[stem7x7 B{B} {H}x{W} s{stride} {'LN' if ln else 'ReLU'}] max |err| / max |ref| {err:.2e}; {1e3 * ms:.1f} us per launch")
    _close(got, ref, 2e-5 if ln else 2e-5 * 60.0, "stem7x7")


@pytest.mark.parametrize("rows,res", [(51200, False), (51200, True), (1600, True), (70, False), (33, True)])
def test_thin128_linear(ops, rows, res):
    """r06 (thin_linear.hip): the 128 -> 128 projections of the MiT stage-2 blocks (q: mix_transformers.py:110; proj + residual: :137-138, :199) in the transposed,
    register-epilogue form.  Oracle: torch fp64; rows with offsets and an outlier channel, ragged last tile, in place on the residual."""
    x = _rand((rows, 128), 70) * 2.0 + 0.3
    x[3] += 40.0
    x[5, 77] = 300.0
    w, b = _rand((128, 128), 71, 1.0 / math.sqrt(128)), 0.1 * _rand((128,), 72)
    r = _rand((rows, 128), 73) if res else None
    ref = x.double() @ w.double().t() + b.double() + (r.double() if res else 0.0)
    got = ops.thin128(x.cuda(), w, b, r.cuda() if res else None)
    _, ms = ops.thin128(x.cuda(), w, b, r.cuda() if res else None, iters=10)
    print(f"[thin128 rows {rows} res {res}] max |err| / max |ref| {float((got.double().cpu() - ref).abs().max() / ref.abs().max()):.2e}; {1e3 * ms:.1f} us per launch")
    _close(got, ref, 2e-5, "thin128")
    if res:
        rin = r.cuda().clone()
        assert torch.equal(ops.thin128(x.cuda(), w, b, rin, inplace=True), got)


def test_sr_attention_split_f16_extremes(ops):
    """peaked rows (|logit| ~ 100), tiny and large K / V magnitudes, V beyond the +-4094 range of the scaled split
    (saturates: finite output)"""
    B, N, heads, M = 1, 64, 1, 100
    q = _rand((B, N, 64), 28)
    kv = _rand((B, M, 128), 29)
    kv[0, 17, :64] = q[0, 5] * 6.0
    q[0, 9] *= 30.0
    kv[0, 40:60, 64:] *= 1e-4
    kv[0, 60:70, 64:] *= 300.0  # up to ~1200: inside the +-4094 range
    ref, bound = _attention_bound(q, kv, heads)
    got = ops.sr_attention_variant(q.cuda(), kv.cuda(), heads, 1).double().cpu()
    assert float(((got - ref).abs() / bound).max()) <= 1.0
    kv[0, 3, 64:] = 50000.0
    assert torch.isfinite(ops.sr_attention_variant(q.cuda(), kv.cuda(), heads, 1)).all()


def test_sr_attention_spiked_row(ops):
    """one key dominating a query row (softmax max path) and a large-magnitude row"""
    B, N, heads, M = 1, 64, 1, 100
    q = _rand((B, N, 64), 28)
    kv = _rand((B, M, 128), 29)
    kv[0, 17, :64] = q[0, 5] * 6.0
    q[0, 9] *= 30.0
    a = ((q.double() @ kv.double()[..., :64].transpose(-2, -1)) * 0.125).softmax(-1)
    ref = a @ kv.double()[..., 64:]
    _close(ops.sr_attention_variant(q.cuda(), kv.cuda(), heads, 0), ref, 1e-5, "spiked attention (exact fp32 MFMA kernel)")


@pytest.mark.parametrize("B,H,W,C", [(2, 10, 10, 256), (1, 80, 80, 256), (1, 160, 160, 64), (1, 3, 5, 8)])
def test_upsample2x(ops, B, H, W, C):
    x = _rand((B, H, W, C), 30)
    ref = F.interpolate(x.double().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1).contiguous()
    _close(ops.upsample2x(x.cuda()), ref, 2e-6, "upsample2x")


def _region_report(got, ref):
    """where a wrong result sits: worst |err| per 32-row x 32-column tile (debug aid for fragment-layout mistakes)"""
    err = (got.double().cpu() - ref.double()).abs()
    R, N = err.shape
    out = []
    for r0 in range(0, min(R, 128), 32):
        out.append(" ".join(f"{float(err[r0:r0 + 32, c0:c0 + 32].max()):8.1e}" for c0 in range(0, min(N, 640), 32)))
    return "\n".join(out)


@pytest.mark.parametrize("K,N,tokens,images", [(320, 320, 400, 3), (320, 640, 100, 5), (320, 1280, 400, 2), (1280, 320, 400, 2), (320, 320, 50, 2), (768, 320, 70, 3)])
def test_rb_linear(ops, K, N, tokens, images):
    """Row-block linear layers (rb_gemm.hip): resident and streamed forms, full and ragged row blocks (400 = 6 x 64 + 16, 100 = 64 + 36, 50, 70 = 64 + 6),
    with / without the staged LayerNorm (rows with offsets and outlier channels), GELU / residual epilogues, in-place residual.  Oracle: torch fp64."""
    rows = tokens * images
    x = _rand((rows, K), 231, 1.5)
    w = _rand((N, K), 233, 1.0 / math.sqrt(K))
    w[5] *= 37.0                                   # rows of very different scale: the per-output-channel power-of-two weight scale
    w[N - 1] *= 1e-3
    b = _rand((N,), 234, 0.1)
    r = _rand((rows, N), 237)
    ref = F.linear(x.double(), w.double(), b.double())
    got = ops.rb_linear(x.cuda(), w, b, tokens)
    try:
        _close(got, ref, 3e-5 * math.sqrt(K / 320), "rb_linear")
    except AssertionError as e:
        raise AssertionError(str(e) + "\n" + _region_report(got, ref))
    if K == 320:
        _close(ops.rb_linear(x.cuda(), w, b, tokens, act=2), pf_oracle.gelu(ref), 3e-5, "rb_linear+gelu")
    else:
        with pytest.raises(Exception):
            ops.rb_linear(x.cuda(), w, b, tokens, act=2)   # the streamed form (fc2) carries no activation: loud
    rc = r.cuda()
    _close(ops.rb_linear(x.cuda(), w, b, tokens, res=rc), ref + r.double(), 3e-5 * math.sqrt(K / 320), "rb_linear+res")
    if K == 320:
        _close(ops.rb_linear(x.cuda(), w, b, tokens, act=2, res=rc), pf_oracle.gelu(ref) + r.double(), 3e-5, "rb_linear+gelu+res")
    if K == 320:
        xo = _with_outlier_channels(x + 10.0 * _rand((rows, 1), 232), 238)
        xo[3] = 0.0                                # a constant row: variance 0
        g = 1 + _rand((K,), 235, 0.3)
        be = _rand((K,), 236, 0.2)
        for eps in (1e-6, 1e-5):
            refl = F.linear(F.layer_norm(xo.double(), (K,), g.double(), be.double(), eps), w.double(), b.double())
            _close(ops.rb_linear(xo.cuda(), w, b, tokens, gamma=g, beta=be, eps=eps), refl, 5e-5, "rb_linear with LayerNorm")


@pytest.mark.parametrize("B,Hr,Wr", [(3, 10, 10), (2, 6, 7), (1, 2, 2)])
def test_rb_srkv(ops, B, Hr, Wr):
    """The key / value branch of a stage-3 MiT block as one launch (rb_chain.hip): LayerNorm-1 of the gathered source tokens, 2 x 2 / stride-2 conv, LayerNorm, kv --
    100 reduced tokens per image (blocks of 32 / 32 / 32 / 4), ragged maps, rows with offsets and outlier channels.  Oracle: torch fp64 (mix_transformers.py:119-127)."""
    C = 320
    x = _with_outlier_channels(_rand((B, 2 * Hr, 2 * Wr, C), 301, 1.5) + 5.0 * _rand((B, 2 * Hr, 2 * Wr, 1), 302), 303)
    g1, b1 = 1 + _rand((C,), 304, 0.3), _rand((C,), 305, 0.2)
    wsr, bsr = _rand((C, C, 2, 2), 306, 1.0 / math.sqrt(4 * C)), _rand((C,), 307, 0.1)
    g2, b2 = 1 + _rand((C,), 308, 0.3), _rand((C,), 309, 0.2)
    wkv, bkv = _rand((2 * C, C), 310, 1.0 / math.sqrt(C)), _rand((2 * C,), 311, 0.1)
    wkv[7] *= 29.0
    xn = F.layer_norm(x.double(), (C,), g1.double(), b1.double(), 1e-6)
    y = F.conv2d(xn.permute(0, 3, 1, 2), wsr.double(), bsr.double(), stride=2).permute(0, 2, 3, 1).reshape(B, Hr * Wr, C)
    ref = F.linear(F.layer_norm(y, (C,), g2.double(), b2.double(), 1e-5), wkv.double(), bkv.double())
    got = ops.rb_srkv(x.cuda(), g1, b1, 1e-6, wsr, bsr, g2, b2, 1e-5, wkv, bkv)
    try:
        _close(got, ref, 6e-5, "rb_srkv")
    except AssertionError as e:
        raise AssertionError(str(e) + "\n" + _region_report(got.reshape(-1, 2 * C), ref.reshape(-1, 2 * C)))


@pytest.mark.parametrize("tokens,images", [(400, 2), (100, 3), (70, 2)])
def test_rb_proj_fc1(ops, tokens, images):
    """x1 = x + proj(attn); hidden = fc1(LayerNorm_2(x1)) as one launch (rb_chain.hip): full and ragged row blocks, residual rows with offsets and outlier channels
    (the LayerNorm runs on the projection's accumulators).  Oracle: torch fp64 (mix_transformers.py:137-139, :199-200, :52)."""
    C = 320
    rows = tokens * images
    attn = _rand((rows, C), 401, 1.2)
    x = _with_outlier_channels(_rand((rows, C), 402, 1.5) + 8.0 * _rand((rows, 1), 403), 404)
    wp, bp = _rand((C, C), 405, 1.0 / math.sqrt(C)), _rand((C,), 406, 0.1)
    g, be = 1 + _rand((C,), 407, 0.3), _rand((C,), 408, 0.2)
    w1, b1 = _rand((4 * C, C), 409, 1.0 / math.sqrt(C)), _rand((4 * C,), 410, 0.1)
    w1[11] *= 23.0
    x1_ref = x.double() + F.linear(attn.double(), wp.double(), bp.double())
    h_ref = F.linear(F.layer_norm(x1_ref, (C,), g.double(), be.double(), 1e-6), w1.double(), b1.double())
    x1, hid = ops.rb_proj_fc1(attn.cuda(), x.cuda(), wp, bp, g, be, 1e-6, w1, b1, tokens)
    _close(x1, x1_ref, 3e-5, "rb_proj_fc1 x1")
    try:
        _close(hid, h_ref, 6e-5, "rb_proj_fc1 hidden")
    except AssertionError as e:
        raise AssertionError(str(e) + "\n" + _region_report(hid, h_ref))


WINO_CASES = [
    # name, B, H, W, Cin, Cout: the four decoder map sizes (80^2 = whole blocks of 16 x 16 pixels; 40^2, 20^2, 10^2: ragged last blocks), odd sizes, other channel counts
    ("rcu80", 1, 80, 80, 256, 256),
    ("rcu40", 2, 40, 40, 256, 256),
    ("rcu20", 2, 20, 20, 256, 256),
    ("rcu10", 3, 10, 10, 256, 256),
    ("odd23x37", 2, 23, 37, 64, 128),
    ("one_tile_row", 1, 1, 33, 128, 64),
    ("cin96_six_chunks", 1, 18, 16, 96, 64),
    # r06: maps with 1 <= H mod 16 <= 8 run wino256x64d in its HALF-PATCH geometry (two 8 x 16 half patches per block: rcu40, rcu20, odd23x37, one_tile_row and
    # cin96_six_chunks above already do); an ODD number of half patches (1 x 5 x 3: the last block's second half is dead) and two images sharing a block
    ("half_odd40", 1, 40, 40, 64, 64),
    ("half_pairs_across_images", 3, 8, 40, 64, 128),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", WINO_CASES, ids=[c[0] for c in WINO_CASES])
def test_winograd_tiles_are_bit_identical(ops, case):
    """The two Winograd kernels differ in how the instructions are scheduled, not in arithmetic or accumulation order."""
    name, B, H, W, Cin, Cout = case
    names = ops.conv_tiles()
    x = _rand((B, H, W, Cin), 310).cuda()
    w = _rand((Cout, Cin, 3, 3), 311, 1.0 / math.sqrt(Cin * 9))
    b = _rand((Cout,), 312, 0.1)
    r1 = _rand((B, H, W, Cout), 313).cuda()
    outs = {t: ops.conv2d(x, w, b, pad=1, act=1, res1=r1, post_relu=True, tile=names.index(t), splitk=False) for t in ("wino256x64c", "wino256x64d")}
    assert torch.equal(outs["wino256x64c"], outs["wino256x64d"]), name


@pytest.mark.parametrize("tile_name", ["wino256x64c", "wino256x64d"])
@pytest.mark.parametrize("case", WINO_CASES, ids=[c[0] for c in WINO_CASES])
def test_conv2d_winograd_tile(ops, case, tile_name):
    """Winograd F(2x2, 3x3) tiles (wino.hip: 16 position GEMMs on the split-f16 MFMA, input / output transforms in fp32) against fp64, with the whole
    epilogue (bias, ReLU, two residuals, ReLU after the residuals -- the ResidualConvUnit forms of decode_head.py:242-256) and without; error measured against the
    natural scale of a dot product, sum |x||w|, and held to 4x the direct halo tile's error + an fp32 floor (the transforms add a few fp32 roundings)."""
    name, B, H, W, Cin, Cout = case
    names = ops.conv_tiles()
    assert tile_name in names
    tw = names.index(tile_name)
    th = names.index("sbh128x64")
    assert ops.conv2d_bench(B, H, W, Cin, Cout, 3, 1, 1, tile=tw, iters=1) > 0   # the tile really runs this shape (an unusable tile id would fall back silently) ...
    # ... and splitk=False below: with the engine's split-K rule on (deep K, few blocks) pf_op_conv2d would hand the small shapes to the linear tiles instead
    x = _rand((B, H, W, Cin), 300)
    w = _rand((Cout, Cin, 3, 3), 301, 1.0 / math.sqrt(Cin * 9))
    b = _rand((Cout,), 302, 0.1)
    r1 = _rand((B, H, W, Cout), 303)
    r2 = _rand((B, H, W, Cout), 304)
    ref0 = _ref_conv(x, w, b, 1, 1)
    scale = _ref_conv(x.abs(), w.abs(), None, 1, 1)
    xd = x.cuda()
    got = ops.conv2d(xd, w, b, pad=1, tile=tw, splitk=False).double().cpu()
    direct = ops.conv2d(xd, w, b, pad=1, tile=th).double().cpu()
    ew, ed = ((got - ref0).abs() / scale).max().item(), ((direct - ref0).abs() / scale).max().item()
    print(f"[{tile_name} {name}] |err| / sum|x||w|: winograd {ew:.2e}, direct halo tile {ed:.2e}")
    assert ew <= 4 * ed + 2.0 ** -20, (name, ew, ed)
    # full epilogue: y = relu(relu(conv + bias) + res1 + res2)
    ref1 = torch.relu(torch.relu(ref0) + r1.double() + r2.double())
    got1 = ops.conv2d(xd, w, b, pad=1, act=1, res1=r1.cuda(), res2=r2.cuda(), post_relu=True, tile=tw, splitk=False).double().cpu()
    assert ((got1 - ref1).abs() / (scale + 1.0)).max().item() <= 4 * ed + 2.0 ** -20, name
    # no bias
    got2 = ops.conv2d(xd, w, None, pad=1, tile=tw, splitk=False).double().cpu()
    assert ((got2 - _ref_conv(x, w, None, 1, 1)).abs() / scale).max().item() <= 4 * ed + 2.0 ** -20, name
