"""The reference's only pinned results (need the trained weights, not downloadable here): runs when
PF_WEIGHTS_DIR holds paramnet_360cities_edina_rpf.pth and PF_ASSETS_DIR the reference's assets/imgs."""
import os

import numpy as np
import pytest

W = os.path.join(os.environ.get("PF_WEIGHTS_DIR", "/nonexistent"), "paramnet_360cities_edina_rpf.pth")
A = os.environ.get("PF_ASSETS_DIR", "/root/reference/assets/imgs")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(W), reason="trained weights not available offline")]


def _load_bgr(path):
    from PIL import Image

    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


@pytest.mark.parametrize("img,roll,pitch,vfov", [("cityscape.jpg", 4.54, 48.88, 52.82), ("epic.png", 20.19, -68.75, 65.36)])
def test_demo_known_answers(img, roll, pitch, vfov):
    """demo/demo.py:145-161 and notebooks/predict_perspective_fields.ipynb:63-66 (2-decimal prints)."""
    from perspectivefields_amd import PerspectiveFields

    m = PerspectiveFields("Paramnet-360Cities-edina-centered", weights=W).eval().cuda()
    p = m.inference(_load_bgr(os.path.join(A, img)))
    assert abs(float(p["pred_roll"]) - roll) < 0.02
    assert abs(float(p["pred_pitch"]) - pitch) < 0.02
    assert abs(float(p["pred_vfov"]) - vfov) < 0.02
