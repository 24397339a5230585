"""N1: device-side PIL resize, bit-identical to Image.resize(BILINEAR) (reference perspectivefields.py:34-46)."""
import numpy as np
import pytest
import torch
from PIL import Image

from perspectivefields_amd.synth import synthetic_image

pytestmark = pytest.mark.gpu
SIZES = [(640, 640), (721, 900), (64, 48), (100, 333), (320, 320), (1080, 1920), (317, 2), (5, 700), (1365, 1024), (2739, 5477)]


@pytest.fixture(scope="module")
def model():
    from perspectivefields_amd import PerspectiveFields

    return PerspectiveFields("Paramnet-360Cities-edina-centered", weights="synthetic:0").eval().cuda()


@pytest.mark.parametrize("hw", SIZES, ids=[f"{h}x{w}" for h, w in SIZES])
def test_device_resize_is_bit_identical_to_pil(model, hw):
    h, w = hw
    img = np.random.default_rng(h * 7919 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((320, 320), Image.BILINEAR))
    out = torch.empty((320, 320, 3), dtype=torch.uint8, device="cuda")
    model._get_engine().resize_into(torch.from_numpy(img).cuda(), out)
    got = out.cpu().numpy()
    assert np.array_equal(got, ref), f"{(got != ref).sum()} differing bytes, max |d| {np.abs(got.astype(int) - ref.astype(int)).max()}"


def test_inference_with_device_resize_matches_host_resize(model):
    imgs = [synthetic_image(480, 640, 5), synthetic_image(700, 500, 6)]
    model.device_resize = False
    a = model.inference_batch(imgs)
    model.device_resize = True
    b = model.inference_batch(imgs)
    model.device_resize = False
    for x, y in zip(a, b):
        assert torch.equal(x["pred_gravity_original"], y["pred_gravity_original"])
        assert torch.equal(x["pred_latitude_original"], y["pred_latitude_original"])
        assert float(x["pred_roll"]) == float(y["pred_roll"])


def test_batched_device_resize_is_bit_identical_to_pil(model):
    """pf_resize_batch_u8: 40 mixed-size images (more than one 32-image launch pair) in one call, byte-identical to PIL."""
    sizes = [(640, 640), (384, 512), (1024, 1365), (64, 48), (317, 2), (5, 700), (320, 320), (721, 900)] * 5
    rng = np.random.default_rng(4242)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    out = torch.empty((len(imgs), 320, 320, 3), dtype=torch.uint8, device="cuda")
    model._get_engine().resize_batch_into([torch.from_numpy(im).cuda() for im in imgs], out)
    got = out.cpu().numpy()
    for i, im in enumerate(imgs):
        ref = np.asarray(Image.fromarray(im).resize((320, 320), Image.BILINEAR))
        assert np.array_equal(got[i], ref), (i, sizes[i], int((got[i] != ref).sum()))
