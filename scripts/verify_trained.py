"""First-contact kit for TRAINED weights (VERDICT r05 "missing #1"): one command, one table.

All parity evidence of this repository is on seeded synthetic checkpoints -- the zoo's .pth files are not downloadable here.  This script is what to run the first
time a real checkpoint is on the box.  It cannot be passed in the build container; it must exist so that the first real checkpoint is a one-command check.

    PF_WEIGHTS_DIR=/path/to/checkpoints python scripts/verify_trained.py [--assets /path/to/reference/assets/imgs] [--versions v1,v2] [--reference /root/reference]

Per zoo version whose checkpoint file is found in PF_WEIGHTS_DIR (file names: perspectivefields_amd.config.model_zoo[version]["weights"]):
  1. range report (PerspectiveFields.check_range -> pf_debug_forward_u8 flags = 2): max |x| over the ~330 dense-layer inputs, smallest rms, layers outside the
     split-f16 window (65504; 16376 in front of a Winograd conv; 8188 / 4094 for attention q / kv), static bound of the unwatched tensors (pf_static_window_max);
  2. what precision="auto" settles on, and why;
  3. the reference's two printed known answers (demo/demo.py:145-161, notebooks/predict_perspective_fields.ipynb:63-66: roll / pitch / vfov of cityscape.jpg and
     epic.png to two decimals; Paramnet-360Cities-edina-centered only);
  4. fast mode (fp32 = split-f16) against the exact mode (fp32_bf16x6) on every asset image: up-vector 1 - cos, latitude L1, ParamNet scalars -- tolerances of
     BASELINE.json (1e-3 / 1e-3 / 1e-4); the saturation counter after the fast run;
  5. where the reference tree is importable (--reference, through oracle/ref_shim.py): the unmodified reference on CPU with the SAME checkpoint against the HIP path.
Exit code 0 only if every check that could run passed.
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KNOWN = {"cityscape.jpg": (4.54, 48.88, 52.82), "epic.png": (20.19, -68.75, 65.36)}   # demo/demo.py:145-161, ipynb:63-66
TOL = {"up_1_minus_cos": 1e-3, "latitude_l1_deg": 1e-3, "paramnet": 1e-4}
SCALARS = ("pred_roll", "pred_pitch", "pred_vfov", "pred_general_vfov", "pred_rel_focal", "pred_rel_cx", "pred_rel_cy")


def load_bgr(path):
    from PIL import Image

    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def deltas(a, b):
    import torch

    g, go = a["pred_gravity_original"].double().cpu(), b["pred_gravity_original"].double().cpu()
    dcos = float((1.0 - (g * go).sum(0) / torch.sqrt((g * g).sum(0) * (go * go).sum(0)).clamp_min(1e-30)).max())
    dlat = float((a["pred_latitude_original"].double().cpu() - b["pred_latitude_original"].double().cpu()).abs().mean())
    dpar = max((abs(float(a[k]) - float(b[k])) for k in SCALARS if k in a and k in b), default=0.0)
    return dcos, dlat, dpar


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--assets", default=os.environ.get("PF_ASSETS_DIR", "/root/reference/assets/imgs"))
    ap.add_argument("--versions", default="")
    ap.add_argument("--reference", default="/root/reference", help="reference tree for check 5 (skipped when absent)")
    ap.add_argument("--max-side", type=int, default=2048, help="asset images larger than this are skipped in check 5 (the CPU reference is slow)")
    args = ap.parse_args(argv)

    import torch

    from perspectivefields_amd import PerspectiveFields
    from perspectivefields_amd.config import model_zoo

    wdir = os.environ.get("PF_WEIGHTS_DIR")
    if not wdir or not os.path.isdir(wdir):
        print("PF_WEIGHTS_DIR is not set (or not a directory): nothing to verify.  Expected files:")
        for v, e in model_zoo.items():
            print(f"  {os.path.basename(e['weights']):48s} <- {v}")
        return 2
    if not torch.cuda.is_available():
        print("no GPU visible: the HIP engine has no CPU path")
        return 2
    images = {os.path.basename(p): load_bgr(p) for p in sorted(glob.glob(os.path.join(args.assets, "*"))) if p.lower().endswith((".jpg", ".jpeg", ".png"))}
    if not images:
        print(f"no images under {args.assets}: using two synthetic ones (the known answers cannot be checked)")
        from perspectivefields_amd.synth import synthetic_image

        images = {"synthetic_640.png": synthetic_image(640, 640, 1), "synthetic_480x720.png": synthetic_image(480, 720, 2)}
    want = [v for v in args.versions.split(",") if v] or list(model_zoo)
    rows, failed = [], False

    def row(version, check, value, ok):
        nonlocal failed
        rows.append((version, check, value, {True: "ok", False: "FAIL", None: "-"}[ok]))
        failed = failed or ok is False

    for version in want:
        path = os.path.join(wdir, os.path.basename(model_zoo[version]["weights"]))
        if not os.path.exists(path):
            row(version, "checkpoint", f"{os.path.basename(path)} not in PF_WEIGHTS_DIR", None)
            continue
        imgs = list(images.values())
        # 1 + 2: range report, what auto settles on
        m_auto = PerspectiveFields(version, weights=path).eval().cuda()
        rep = m_auto.check_range(imgs[:4], verbose=False)
        layers = rep["layers"]
        row(version, "dense-layer inputs: max |x| / smallest rms", f"{max(r['max_abs'] for r in layers):.4g} / {min(r['rms'] for r in layers if r['rms'] > 0):.3g} over {len(layers)} tensors", None)
        wino = [r for r in layers if "[winograd]" in r["name"]]
        if wino:
            row(version, "Winograd-layer inputs: max |x| (window 16376)", f"{max(r['max_abs'] for r in wino):.4g}", max(r["max_abs"] for r in wino) <= 16376.0)
        row(version, "layers outside the split-f16 window", f"{len(rep['saturated'])} saturated, {len(rep['tiny'])} all-tiny, {len(rep['non_finite'])} non-finite"
            + (f"; first: {(rep['saturated'] + rep['tiny'] + rep['non_finite'])[0]['name']}" if not rep["ok"] else ""), rep["ok"])
        row(version, "static bound of unwatched tensors (<= 65504)", f"{m_auto._get_engine().static_window_max():.4g}", m_auto._get_engine().static_window_max() <= 65504.0)
        out_auto = m_auto.inference_batch(imgs)
        row(version, "precision='auto' settles on", f"{m_auto.precision}: {getattr(m_auto, 'precision_reason', '')[:110]}", None)
        # 3: the reference's printed known answers
        if version == "Paramnet-360Cities-edina-centered":
            for name, (roll, pitch, vfov) in KNOWN.items():
                if name in images:
                    p = out_auto[list(images).index(name)]
                    got = (float(p["pred_roll"]), float(p["pred_pitch"]), float(p["pred_vfov"]))
                    row(version, f"known answer {name} (roll, pitch, vfov)", f"{got[0]:.2f} {got[1]:.2f} {got[2]:.2f}  (reference prints {roll:.2f} {pitch:.2f} {vfov:.2f})",
                        all(abs(g - w) < 0.02 for g, w in zip(got, (roll, pitch, vfov))))
        # 4: fast mode against the exact mode, image by image
        m_fast = PerspectiveFields(version, weights=path, precision="fp32").eval().cuda()
        m_exact = PerspectiveFields(version, weights=path, precision="fp32_bf16x6").eval().cuda()
        eng = m_fast._get_engine()
        before = int(eng.saturation_snapshot())
        worst = [0.0, 0.0, 0.0]
        for im in imgs:
            d = deltas(m_fast.inference(im), m_exact.inference(im))
            worst = [max(a, b) for a, b in zip(worst, d)]
        moved = int(eng.saturation_snapshot()) - before
        row(version, f"fp32 (split-f16) vs fp32_bf16x6 on {len(imgs)} images: 1-cos / lat L1 / scalars", f"{worst[0]:.2e} / {worst[1]:.2e} deg / {worst[2]:.2e}",
            worst[0] <= TOL["up_1_minus_cos"] and worst[1] <= TOL["latitude_l1_deg"] and (worst[2] <= TOL["paramnet"] or not m_fast.param_on))
        row(version, "saturation counter moved by the fast-mode run", str(moved), moved == 0)
        # 5: the unmodified reference on CPU, same checkpoint
        try:
            sys.path.insert(0, ROOT)
            from oracle import ref_shim

            if os.path.isdir(args.reference) and ref_shim.reference_available():
                sd = torch.load(path, map_location="cpu", weights_only=True)
                ref_model = ref_shim.build_reference(version, sd["model"] if "model" in sd else sd)
                small = [(n, im) for n, im in images.items() if max(im.shape[:2]) <= args.max_side][:3]
                worst = [0.0, 0.0, 0.0]
                with torch.no_grad():
                    for n, im in small:
                        worst = [max(a, b) for a, b in zip(worst, deltas(m_exact.inference(im), ref_model.inference(im)))]
                row(version, f"HIP (exact mode) vs the unmodified reference on CPU, {len(small)} images", f"{worst[0]:.2e} / {worst[1]:.2e} deg / {worst[2]:.2e}",
                    worst[0] <= TOL["up_1_minus_cos"] and worst[1] <= TOL["latitude_l1_deg"] and (worst[2] <= TOL["paramnet"] or not m_exact.param_on))
            else:
                row(version, "HIP vs the unmodified reference", f"reference tree not importable at {args.reference}", None)
        except Exception as e:  # the shim failing must not hide the other checks
            row(version, "HIP vs the unmodified reference", f"skipped: {e!r}"[:120], None)
        del m_auto, m_fast, m_exact

    w0 = max(len(r[0]) for r in rows)
    w1 = max(len(r[1]) for r in rows)
    print(f"{'version':{w0}s}  {'check':{w1}s}  result")
    for v, c, val, ok in rows:
        print(f"{v:{w0}s}  {c:{w1}s}  [{ok:>4s}] {val}")
    print("\nALL CHECKS THAT RAN PASSED" if not failed else "\nAT LEAST ONE CHECK FAILED: use precision='auto' (default) or 'fp32_bf16x6' for that checkpoint and send the table upstream")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
