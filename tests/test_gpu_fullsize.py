"""Headline-configuration parity (-m gpu): the sizes and batch BASELINE.json names -- 640x640 (configs[0..3]),
384x512 / 1024x1365 (configs[4]), batch 32 (configs[2]) -- against
  (a) golden vectors of the UNMODIFIED reference at those sizes (tests/golden/fullsize.npz, oracle/gen_golden.py),
  (b) the CPU oracle's post-process applied to the HIP path's own 320x320 predictions, compared at EVERY output pixel
      (the up-sampling regime of gravity_head.py:248-257 / utils/utils.py:503-506: source index clamped at 0 on the
      first rows / columns, neighbour clamped on the last),
  (c) the CPU oracle run live on images 0 / 15 / 31 of a batch of 32, and the same images through a batch of 1.
Tolerances are BASELINE.json's (tests/parity.py); everything goes through the C ABI (perspectivefields_amd.engine)."""
import os

import numpy as np
import pytest
import torch

from oracle import pf_oracle
from perspectivefields_amd.config import arch_of, get_cfg
from perspectivefields_amd.synth import synthetic_image, synthetic_state_dict, to_torch
from tests.parity import TOL_COS, TOL_LAT_L1, TOL_PARAM, assert_fields_close, l1, one_minus_cos

pytestmark = pytest.mark.gpu

CASES = {
    "centered": "Paramnet-360Cities-edina-centered",
    "persnet": "PersNet-360Cities",
    "uncentered": "Paramnet-360Cities-edina-uncentered",
}
_models = {}
_oracle_cache = {}


PRECISIONS = ["fp32", "fp32_bf16x6"]   # the windowed default (2-way fp16 split) and the exact bf16 split `precision="auto"` falls to outside the window: same tolerances


def model(tag, precision=None):
    key = tag if precision is None else (tag, precision)
    if key not in _models:
        from perspectivefields_amd import PerspectiveFields

        _models[key] = PerspectiveFields(CASES[tag], weights="synthetic:0", **({} if precision is None else {"precision": precision})).eval().cuda()
    return _models[key]


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("tag", ["centered", "uncentered", "persnet"])
def test_fullsize_vs_reference_golden(tag, precision, golden_dir):
    g = np.load(os.path.join(golden_dir, "fullsize.npz"))
    pre = f"fs_{tag}_"
    n = int(g[pre + "n"])
    m = model(tag, precision)
    assert m._get_engine().precision == precision
    batched = []
    for k in range(n):
        H, W = (int(v) for v in g[f"{pre}size_{k}"])
        batched.append({"image": torch.as_tensor(g[f"{pre}in_u8_{k}"].astype("float32").transpose(2, 0, 1)), "height": H, "width": W})
    res = m.forward(batched)
    names = [str(x) for x in g[pre + "param_names"]]
    cls = tag == "persnet"
    for k, r in enumerate(res):
        H, W = (int(v) for v in g[f"{pre}size_{k}"])
        up, lat = r["pred_gravity_original"].cpu().numpy(), r["pred_latitude_original"].cpu().numpy()
        assert up.shape == (2, H, W) and lat.shape == (H, W)
        rows, cols = [0, 1, H - 2, H - 1], [0, 1, W - 2, W - 1]
        parts = {
            "s8": (up[:, ::8, ::8], g[f"{pre}grav_s8_{k}"], lat[::8, ::8], g[f"{pre}lat_s8_{k}"]),
            "border rows": (up[:, rows, :], g[f"{pre}grav_rows_{k}"], lat[rows, :], g[f"{pre}lat_rows_{k}"]),
            "border cols": (up[:, :, cols], g[f"{pre}grav_cols_{k}"], lat[:, cols], g[f"{pre}lat_cols_{k}"]),
        }
        for what, (a, a_ref, b, b_ref) in parts.items():
            assert a.shape == a_ref.shape and b.shape == b_ref.shape
            if cls:  # argmax decode: a flipped bin moves a pixel by a whole bin -- bound the fraction of such pixels
                assert np.mean(one_minus_cos(a, a_ref) > TOL_COS) <= 5e-3, (tag, k, what)
                assert np.mean(np.abs(b - b_ref) > 1e-3) <= 5e-3, (tag, k, what)
            else:
                c, e = assert_fields_close(a, a_ref, b, b_ref, f"{tag} img{k} {H}x{W} {what}")
                print(f"[fullsize {tag} {precision} img{k} {H}x{W} {what}] 1-cos max {c:.2e}  latitude L1 {e:.2e} deg")
        if cls:
            fg = float((r["pred_gravity"].argmax(0).cpu().numpy() != g[f"{pre}grav_argmax_{k}"]).mean())
            fl = float((r["pred_latitude"].argmax(0).cpu().numpy() != g[f"{pre}lat_argmax_{k}"]).mean())
            print(f"[fullsize persnet img{k}] argmax mismatch fraction gravity {fg:.2e} latitude {fl:.2e}")
            assert fg <= 2e-3 and fl <= 2e-3
        else:
            pg, pl = r["pred_gravity"].cpu().numpy(), r["pred_latitude"].cpu().numpy()
            assert_fields_close(pg[:, ::2, ::2], g[f"{pre}grav_s2_{k}"], pl[:, ::2, ::2], g[f"{pre}lat_s2_{k}"], f"{tag} img{k} 320^2")
        if names:
            d = np.abs(np.array([float(r[nm]) for nm in names]) - g[f"{pre}params_{k}"])
            print(f"[fullsize {tag} {precision} img{k}] ParamNet max|d| {d.max():.2e}")
            assert d.max() <= TOL_PARAM, dict(zip(names, d))


SIZES_DENSE = [(640, 640), (384, 512), (1024, 1365), (641, 479), (200, 150), (320, 320), (321, 319)]


@pytest.mark.parametrize("tag", ["centered", "persnet"])
def test_postprocess_every_pixel_vs_oracle(tag):
    """pf_postprocess / pf_postprocess_batch at the BASELINE sizes, every output pixel including the 1-px borders,
    against the oracle's restatement of GravityDecoder.postprocess / LatitudeDecoder.postprocess applied to the SAME
    320x320 predictions (so only the post-process arithmetic is compared)."""
    m = model(tag)
    eng = m._get_engine()
    arch = arch_of(get_cfg(CASES[tag]))
    imgs = [synthetic_image(96, 128, seed=700 + i) for i in range(len(SIZES_DENSE))]
    x = torch.from_numpy(np.stack([m.aug.apply_image(im) for im in imgs])).cuda()
    pg, pl, _ = eng.forward(x)
    outs = eng.postprocess_batch(pg, pl, SIZES_DENSE)
    for i, (H, W) in enumerate(SIZES_DENSE):
        gi, li = pg[i].cpu(), pl[i].cpu()
        with torch.no_grad():
            up_ref = pf_oracle.postprocess_gravity(gi, H, W, arch["gravity_cls"], arch["gravity_out"]).numpy()
            lat_ref = pf_oracle.postprocess_latitude(li, H, W, arch["latitude_cls"], arch["latitude_out"]).numpy()
        for what, (up, lat) in {"batch": outs[i], "single": eng.postprocess(pg[i], pl[i], H, W)}.items():
            up, lat = up.cpu().numpy(), lat.cpu().numpy()
            assert up.shape == (2, H, W) and lat.shape == (H, W)
            c = one_minus_cos(up, up_ref)
            d = np.abs(lat - lat_ref)
            # asin is ill-conditioned at |sin| -> 1 (d asin = dv / sqrt(1 - v^2)): one fp32 ulp there is ~0.03 deg
            print(f"[post {tag} {H}x{W} {what}] 1-cos max {c.max():.2e}  |lat| max {d.max():.2e} mean {d.mean():.2e} deg")
            assert c.max() <= 1e-6 and d.mean() <= 1e-4 and d.max() <= 5e-2, (tag, H, W, what)
            # the borders explicitly (clamp branches)
            for sl in (np.s_[0, :], np.s_[-1, :], np.s_[:, 0], np.s_[:, -1]):
                assert np.abs(lat[sl] - lat_ref[sl]).mean() <= 1e-4
                assert one_minus_cos(up[(slice(None),) + sl][:, None], up_ref[(slice(None),) + sl][:, None]).max() <= 1e-6


@pytest.mark.parametrize("precision", PRECISIONS)
def test_batch32_vs_oracle_and_single(precision):
    """BASELINE configs[2]: a batch of 32 640x640 images (the bench batch, with its own tile choices).  Images 0 / 15 /
    31 against the CPU oracle run on the same images, and against the same images run as batches of 1."""
    tag = "centered"
    m = model(tag, precision)
    imgs = [synthetic_image(640, 640, seed=1000 + i) for i in range(32)]
    res = m.inference_batch(imgs)
    assert len(res) == 32
    pick = [0, 15, 31]
    arch = arch_of(get_cfg(CASES[tag]))
    sd = to_torch(synthetic_state_dict(CASES[tag], 0))
    if "b32" not in _oracle_cache:   # one CPU oracle run for both precisions
        with torch.no_grad():
            _oracle_cache["b32"] = pf_oracle.inference_batch(sd, arch, [imgs[i] for i in pick])
    ref = _oracle_cache["b32"]
    for i, o in zip(pick, ref):
        r = res[i]
        c, e = assert_fields_close(r["pred_gravity_original"].cpu().numpy(), o["pred_gravity_original"].numpy(),
                                   r["pred_latitude_original"].cpu().numpy(), o["pred_latitude_original"].numpy(), f"B=32 img{i} 640x640 vs oracle")
        assert_fields_close(r["pred_gravity"].cpu().numpy(), o["pred_gravity"].numpy(), r["pred_latitude"].cpu().numpy(), o["pred_latitude"].numpy(), f"B=32 img{i} 320^2")
        d = max(abs(float(r[k]) - float(o[k])) for k in ("pred_roll", "pred_pitch", "pred_vfov", "pred_rel_focal"))
        print(f"[B=32 {precision} img{i} vs oracle] 1-cos max {c:.2e}  latitude L1 {e:.2e} deg  ParamNet max|d| {d:.2e}")
        assert d <= TOL_PARAM
        s = m.inference(imgs[i])
        c1 = one_minus_cos(r["pred_gravity_original"].cpu().numpy(), s["pred_gravity_original"].cpu().numpy()).max()
        e1 = l1(r["pred_latitude_original"].cpu().numpy(), s["pred_latitude_original"].cpu().numpy())
        d1 = max(abs(float(r[k]) - float(s[k])) for k in ("pred_roll", "pred_pitch", "pred_vfov", "pred_rel_focal"))
        print(f"[B=32 {precision} img{i} vs B=1] 1-cos max {c1:.2e}  latitude L1 {e1:.2e} deg  ParamNet max|d| {d1:.2e}")
        assert c1 <= 1e-6 and e1 <= 2e-4 and d1 <= 5e-5


def test_oversized_batch_is_chunked_or_refused():
    """The kernels address an activation with 32-bit byte offsets: one forward takes at most PF_MAX_BATCH (81) images.
    The C ABI refuses more (PF_ERR_ARG, no silent corruption); the host layer splits longer lists, so an
    inference_batch of 100 images returns what per-image inference returns."""
    from perspectivefields_amd.engine import PfError

    m = model("centered")
    eng = m._get_engine()
    assert eng.max_batch == 81
    too_many = torch.zeros((eng.max_batch + 1, 320, 320, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(PfError, match="PF_MAX_BATCH"):
        eng.forward(too_many)
    # straight through the C ABI as well (bypassing the Python-side check)
    B = eng.max_batch + 1
    pg = torch.empty((B, 2, 320, 320), device="cuda")
    pl = torch.empty((B, 1, 320, 320), device="cuda")
    pp = torch.empty((B, 8), device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    rc = eng.lib.pf_forward_u8(eng._h, B, too_many.data_ptr(), pg.data_ptr(), pl.data_ptr(), pp.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b"PF_MAX_BATCH" in eng.lib.pf_last_error(eng._h)
    assert eng.workspace_bytes(B) == 0
    del too_many, pg, pl, pp, ws
    imgs = [synthetic_image(48, 64, seed=2000 + i) for i in range(100)]
    res = m.inference_batch(imgs)  # 64 + 36
    assert len(res) == 100
    for i in (0, 63, 64, 99):
        s = m.inference(imgs[i])
        assert res[i]["pred_gravity_original"].shape == (2, 48, 64)
        c = one_minus_cos(res[i]["pred_gravity_original"].cpu().numpy(), s["pred_gravity_original"].cpu().numpy()).max()
        e = l1(res[i]["pred_latitude_original"].cpu().numpy(), s["pred_latitude_original"].cpu().numpy())
        d = abs(float(res[i]["pred_roll"]) - float(s["pred_roll"]))
        assert c <= 1e-6 and e <= 2e-4 and d <= 5e-5, (i, c, e, d)
