"""Round-6 GPU tests (-m gpu): the window contract of a model pinned to the fast mode, the race-free / branch-covering saturation watch of the pipelined path,
the device-side general_vfov -> focal of the uncentered models, and the product's sharded entry point on one GPU."""
import warnings

import numpy as np
import pytest
import torch

from oracle import pf_oracle
from perspectivefields_amd.config import arch_of, get_cfg
from perspectivefields_amd.synth import synthetic_image, synthetic_state_dict, to_torch
from tests.parity import l1, one_minus_cos

pytestmark = pytest.mark.gpu

CENTERED = "Paramnet-360Cities-edina-centered"
UNCENTERED = "Paramnet-360Cities-edina-uncentered"
KEYS = ("pred_roll", "pred_pitch", "pred_vfov", "pred_rel_focal")


def _pf(version, sd=None, precision="auto"):
    from perspectivefields_amd import PerspectiveFields

    return PerspectiveFields(version, weights="synthetic:0" if sd is None else sd, precision=precision).eval().cuda()


def _vs(out, ref):
    g, go = out["pred_gravity_original"].double().cpu(), ref["pred_gravity_original"].double().cpu()
    dcos = float((1.0 - (g * go).sum(0) / torch.sqrt((g * g).sum(0) * (go * go).sum(0))).max())
    dlat = float((out["pred_latitude_original"].double().cpu() - ref["pred_latitude_original"].double().cpu()).abs().mean())
    dpar = max(abs(float(out[q]) - float(ref[q])) for q in KEYS if q in out)
    return dcos, dlat, dpar


@pytest.mark.parametrize("target", [30000.0, 250000.0])
def test_pinned_fp32_window_contract(target):
    """include/pf_hip.h pf_set_saturation_counter, "THE CONTRACT FOR A CALLER THAT PINS PF_PRECISION_FP32" (reference layers concerned: the ResidualConvUnit convs,
    decode_head.py:224-256 behind gravity_head.py:139-176).  The first 256 -> 256 conv of the gravity decoder at 80 x 80 runs as Winograd F(2x2, 3x3): its input window
    ends at 65504 / 4 = 16376 and its split does not clamp.  Its input (the folded linear_c1 / linear_c1_proc conv's output) is scaled to `target`: inside (16376, 65504]
    and beyond 65504.  Pinned fp32: the counter MOVES in both cases (the output is then not promised to be finite or right: printed); `auto` on the same checkpoint runs
    in the exact mode and agrees with the oracle."""
    img = synthetic_image(96, 128, seed=11)
    sd = synthetic_state_dict(CENTERED, 0)
    _, _, rng = _pf(CENTERED, sd, "fp32").debug_forward([img], shadow=False, ranges=True)
    # the 256 -> 256 convs of the gravity head at 80 x 80 (M = 6400 at batch 1, K = 9 x 256), in launch order: the first is fusion1.resConfUnit1.conv1, whose input is
    # relu(linear_c1_proc(linear_c1(c1))) -- the output of the folded first conv
    pick = lambda records: [r for r in records if "[winograd]" in r["name"] and "[gravity]" in r["name"] and "M=6400 " in r["name"] and "K=2304" in r["name"] and r["name"].endswith(" x")]
    wino = pick(rng)
    assert wino, [r["name"] for r in rng if "[winograd]" in r["name"]][:8]
    a0 = wino[0]["max_abs"]
    assert 0.0 < a0 < 16376.0, a0
    f = np.float32(target / a0)
    sd2 = dict(sd)
    for k in ("persformer_heads.gravity_head.linear_c1_proc.weight", "persformer_heads.gravity_head.linear_c1_proc.bias"):
        sd2[k] = sd[k] * f      # the conv is linear in (weight, bias): its output scales by f exactly
    mp = _pf(CENTERED, sd2, "fp32")
    eng = mp._get_engine()
    before = int(eng.saturation_snapshot())
    out = mp.inference_batch([img])[0]
    moved = int(eng.saturation_snapshot()) - before
    finite = bool(torch.isfinite(out["pred_gravity_original"]).all()) and bool(torch.isfinite(out["pred_latitude_original"]).all())
    _, _, rng2 = mp.debug_forward([img], shadow=False, ranges=True)
    w2 = pick(rng2)[0]
    print(f"[window contract] Winograd input max |x| {a0:.4g} -> {w2['max_abs']:.4g} (window 16376): pinned fp32 counter +{moved}, outputs finite: {finite}")
    assert w2["max_abs"] > 16376.0
    assert moved > 0, "a pinned-fp32 forward that leaves a Winograd layer's window must move the saturation counter"
    # `auto`: the first batch's range probe sees the tensor beyond its window, the model runs in the exact mode, results at oracle level
    ma = _pf(CENTERED, sd2, "auto")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = ma.inference_batch([img])[0]
    assert ma.precision == "fp32_bf16x6", ma.precision_reason
    with torch.no_grad():
        ref = pf_oracle.inference_batch(to_torch(sd2), arch_of(get_cfg(CENTERED)), [img])[0]
    dcos, dlat, dpar = _vs(got, ref)
    print(f"[window contract] auto -> exact mode vs oracle: up 1-cos {dcos:.2e}  latitude L1 {dlat:.2e} deg  ParamNet max|d| {dpar:.2e}")
    assert dcos <= 1e-3 and dlat <= 1e-3 and dpar <= 1e-3   # (1e-3 on the scalars as in test_auto_watches_every_later_batch: the fp32 oracle itself is ill-conditioned at these scales)


def _ll_scaled_checkpoint():
    """the checkpoint of tests/test_gpu_debug.py::test_auto_watches_every_later_batch: a flat image stays at half the window of conv_fuse_conv0's second input, a
    high-contrast one leaves it"""
    flat = np.full((96, 128, 3), 118, dtype=np.uint8)
    noisy = np.random.default_rng(5).integers(0, 256, (96, 128, 3), dtype=np.uint8)
    sd = synthetic_state_dict(CENTERED, 0)
    _, _, rng = _pf(CENTERED, sd, "fp32").debug_forward([flat], shadow=False, ranges=True)
    a = max(r["max_abs"] for r in rng if r["name"].endswith(" x2"))
    sd2 = dict(sd)
    sd2["ll_enc.conv1.weight"] = sd["ll_enc.conv1.weight"] * np.float32(0.5 * 65504.0 / a)
    return sd2, flat, noisy


def test_stream_rerun_reads_finished_parameters_at_gpu_bound_batches():
    """with the deferred ParamNet branch on, the re-run of a batch that left the window built its scalar entries while the branch was still
    writing them -- invisible at batch 1 (host slower than GPU), a race at GPU-bound batch sizes.  Batches of 48: every result of the pipelined `auto` model (fast mode ->
    window exit in batch 2 -> re-runs) must equal what a model pinned to the exact mode returns for the same batch."""
    sd2, flat, noisy = _ll_scaled_checkpoint()
    B = 48
    batches = [[flat] * B, [flat] * B, [noisy] * (B - 1) + [flat], [flat] * B, [flat] * B]
    exact = _pf(CENTERED, sd2, "fp32_bf16x6")
    want = [exact.inference_batch(b) for b in batches[:3]]
    m = _pf(CENTERED, sd2, "auto")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = list(m.inference_stream(batches, to_host=False, depth=2))
    assert m.precision == "fp32_bf16x6" and any("left the split-f16 window" in str(x.message) for x in w)
    assert len(got) == len(batches) and all(len(g) == B for g in got)
    worst = 0.0
    for bi, wb in ((0, want[0]), (1, want[1]), (2, want[2]), (3, want[0]), (4, want[0])):
        for i in (0, 1, B // 2, B - 2, B - 1):
            dcos, dlat, dpar = _vs(got[bi][i], wb[i])
            worst = max(worst, dpar)
            # batches 0 / 1 ran in the fast mode (in window): oracle-level agreement with the exact mode; 2.. are exact-mode results themselves
            assert dcos <= 1e-3 and dlat <= 1e-3 and dpar <= (1e-3 if bi < 2 else 1e-5), (bi, i, dcos, dlat, dpar)
    print(f"[stream re-run, B = {B}] worst ParamNet |d| vs the pinned exact model over 25 checked images: {worst:.2e}")


def test_stream_watch_covers_the_deferred_paramnet_branch():
    """the snapshot behind forward i was taken BEFORE the ParamNet branch of forward i ran (it runs beside forward i + 1), so an increment by the
    branch was charged to the next batch -- or to nobody for the last one.  The branch cannot be driven out of its window by an image (its input is the normalised
    fields), so the increment is injected: the counter is bumped on the stream that is behind the branch of the LAST batch.  The stream must re-run that batch in
    the exact mode; without the snapshot behind the branch nothing would look at the counter again."""
    m = _pf(CENTERED, None, "auto")
    eng = m._get_engine()
    imgs = [[synthetic_image(80, 96, seed=300 + 4 * b + i) for i in range(4)] for b in range(3)]
    m.inference_batch(imgs[0])          # settles on the fast mode
    assert m.precision == "fp32"
    calls = {"n": 0}
    real = eng.params_ready_event

    def bumped(stream):
        ev = real(stream)
        calls["n"] += 1
        if calls["n"] == len(imgs):     # behind the LAST batch's branch
            with torch.cuda.stream(stream):
                eng._sat.add_(1)
        return ev

    eng.params_ready_event = bumped
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = list(m.inference_stream(imgs, to_host=False, depth=2))
    finally:
        eng.params_ready_event = real
    assert calls["n"] == len(imgs)
    assert m.precision == "fp32_bf16x6" and any("left the split-f16 window" in str(x.message) for x in w), m.precision
    want = _pf(CENTERED, None, "fp32_bf16x6").inference_batch(imgs[-1])
    for g, wv in zip(got[-1], want):
        assert torch.equal(g["pred_gravity"], wv["pred_gravity"]) and all(float(g[k]) == float(wv[k]) for k in KEYS)


def test_uncentered_focal_is_computed_on_the_device():
    """SURVEY row N2 / param_network.py:211-220, utils/utils.py:47-91: pred_rel_focal of ParamNetConvNextRegress comes out of paramnet_scalars_kernel (closed form, fp64)
    -- equal to the host closed form on the same outputs, to the reference's fsolve through the goldens (test_regression_vs_golden[uncentered]), and _param_dicts makes no
    host round trip."""
    from perspectivefields_amd.perspectivefields import general_vfov_to_focal

    m = _pf(UNCENTERED, None, "fp32")
    imgs = [synthetic_image(90 + 7 * i, 120, seed=40 + i) for i in range(5)]
    res, params = m.inference_batch_with_params(imgs)
    assert tuple(params.shape) == (5, 8)
    p = params.double().cpu().numpy()
    want = general_vfov_to_focal(p[:, 3], p[:, 4], (params[:, 2] * 90.0).double().cpu().numpy())
    got = np.array([float(r["pred_rel_focal"]) for r in res])
    print(f"[device focal] rel_focal {got.round(4).tolist()}  max |device - host closed form| {np.abs(got - want).max():.2e}")
    assert np.all(np.isfinite(got)) and np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    assert list(res[0])[5:] == ["pred_roll", "pred_pitch", "pred_general_vfov", "pred_rel_cx", "pred_rel_cy", "pred_rel_focal"]
    # no .cpu() on the way: the entries are views of the device tensor
    orig = torch.Tensor.cpu
    hits = []
    torch.Tensor.cpu = lambda self, *a, **k: (hits.append(1), orig(self, *a, **k))[1]
    try:
        m._param_dicts(params)
    finally:
        torch.Tensor.cpu = orig
    assert not hits, "the uncentered ParamNet entries must not synchronise with the host"


def test_sharded_product_class_on_one_gpu():
    """dist.ShardedPerspectiveFields without a process group (world 1) on the real engine: the three call shapes return what the plain model returns."""
    from perspectivefields_amd.dist import ShardedPerspectiveFields

    m = _pf(CENTERED, None, "fp32")
    spf = ShardedPerspectiveFields(m)
    sizes = [(96, 128), (128, 128), (128, 128), (200, 266)] * 2
    imgs = [synthetic_image(h, w, seed=500 + i) for i, (h, w) in enumerate(sizes)]
    want = m.inference_batch(imgs)
    out = spf.inference_batch(imgs, bucketed=True)
    assert out.indices == list(range(8)) and tuple(out.params.shape) == (8, 8)
    for g, w in zip(out.results, want):
        assert torch.equal(g["pred_gravity_original"], w["pred_gravity_original"]) and all(float(g[k]) == float(w[k]) for k in KEYS)
    assert all(float(out.params[i, 0]) == float(want[i]["pred_roll"]) for i in range(8))
    # the stream: two global batches, device resize on (bit-identical bytes), fields to pinned host memory
    s_out = list(spf.inference_stream([imgs[:4], imgs[4:]], bucketed=True, to_host=True, depth=2))
    assert [o.indices for o in s_out] == [[0, 1, 2, 3]] * 2
    for o, half in zip(s_out, (imgs[:4], imgs[4:])):
        want4 = m.inference_batch(half)     # same batch composition: bit-identical (tile choices depend on the batch size)
        for j, r in enumerate(o.results):
            assert r["pred_gravity_original"].device.type == "cpu"
            assert torch.equal(r["pred_gravity_original"], want4[j]["pred_gravity_original"].cpu())
            assert float(o.params[j, 0]) == float(want4[j]["pred_roll"])
    assert m.device_resize is False
    # the device-resident step of bench.py: pipeline on -> rows one step late, drain() -> the last step's
    eng = m._get_engine()
    u8 = torch.from_numpy(np.stack([m.aug.apply_image(im) for im in imgs])).cuda()
    pg, pl, pr = eng.forward(u8)
    spf.set_pipeline(True)
    o1 = spf.forward_step(u8, sizes)
    o2 = spf.forward_step(u8, sizes)
    last = spf.drain()
    spf.set_pipeline(False)
    assert o1.gathered is None and torch.equal(o2.gathered, pr) and torch.equal(last, pr)
    assert torch.equal(o2.pred_gravity, pg) and torch.equal(o2.fields[3][1], want[3]["pred_latitude_original"])


def test_stage3_batch_split_is_bit_identical(monkeypatch):
    """PF_S3_SPLIT (engine.hip mit(), "the stage-3 split"): MiT stage 3 (mix_transformers.py:198-202, 18 blocks) walks the batch as two half-batches on two streams.
    Same kernels on the same images -- every output of a batch-32 forward must be bit-identical to the one-stream walk, with the deferred ParamNet branch beside it too."""
    from perspectivefields_amd import PerspectiveFields

    x = torch.from_numpy(np.stack([synthetic_image(320, 320, seed=900 + i) for i in range(32)])).cuda()
    outs = {}
    for mode in ("0", "2"):   # 2: split whenever the whole batch passes the row-block gate (the default, 1, waits for a batch whose HALVES pass it: 64)
        monkeypatch.setenv("PF_S3_SPLIT", mode)
        m = PerspectiveFields(CENTERED, weights="synthetic:0", precision="fp32").eval().cuda()
        eng = m._get_engine()
        pg, pl, pr = eng.forward(x)
        eng.set_defer_params(True)
        runs = [eng.forward(x) for _ in range(3)]
        eng.set_defer_params(False)
        torch.cuda.synchronize()
        for g2, l2, p2 in runs:
            assert torch.equal(g2, pg) and torch.equal(l2, pl) and torch.equal(p2, pr), mode
        outs[mode] = (pg, pl, pr)
    for a, b in zip(outs["0"], outs["2"]):
        assert torch.equal(a, b)
