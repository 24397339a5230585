"""Live pin of the oracle and of the checkpoint schema against the UNMODIFIED reference, run only where
/root/reference is importable (the build container; auto-skips on the GPU box)."""
import numpy as np
import pytest
import torch

from oracle import pf_oracle, ref_shim
from perspectivefields_amd.config import arch_of, get_cfg
from perspectivefields_amd.schema import checkpoint_schema
from perspectivefields_amd.synth import synthetic_image, synthetic_state_dict, to_torch
from tests.parity import TOL_PARAM, assert_fields_close

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not present")


@pytest.mark.parametrize("version", ["Paramnet-360Cities-edina-centered", "PersNet-360Cities", "PersNet_Paramnet-GSV-uncentered"])
def test_schema_matches_reference_state_dict(version):
    sd = to_torch(synthetic_state_dict(version, 3))
    ref = ref_shim.build_reference(version, sd)
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    got = {k: tuple(v) for k, v in checkpoint_schema(version).items()}
    assert want == got


def test_oracle_matches_live_reference_other_seed():
    """different weights (seed 5) and image sizes than the golden fixtures"""
    version = "Paramnet-360Cities-edina-centered"
    sd = to_torch(synthetic_state_dict(version, 5))
    ref = ref_shim.build_reference(version, sd)
    imgs = [synthetic_image(70, 110, seed=21), synthetic_image(33, 47, seed=22, smooth=False)]
    with torch.no_grad():
        want = ref.inference_batch(imgs)
        got = pf_oracle.inference_batch(sd, arch_of(get_cfg(version)), imgs)
    for w, g in zip(want, got):
        assert_fields_close(g["pred_gravity"].numpy(), w["pred_gravity"].numpy(), g["pred_latitude"].numpy(), w["pred_latitude"].numpy())
        assert_fields_close(g["pred_gravity_original"].numpy(), w["pred_gravity_original"].numpy(),
                            g["pred_latitude_original"].numpy(), w["pred_latitude_original"].numpy())
        for k in ("pred_roll", "pred_pitch", "pred_vfov", "pred_rel_focal", "pred_general_vfov"):
            assert abs(float(w[k]) - float(g[k])) <= TOL_PARAM
        assert [k for k in w.keys()] == [k for k in g.keys() if not k.startswith("_")]
