"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import shim that lets the *unmodified* reference at /root/reference be imported
in this container, where several of its import-time dependencies (timm, yacs,
omegaconf, cv2, torchvision, equilib, imageio) are not installed.  None of the
stubbed modules contributes arithmetic to the inference path:

  * timm.models.layers: DropPath (identity in eval), to_2tuple, trunc_normal_
    -- used by mix_transformers.py:11 and convnext.py:13 for init only;
  * yacs.config.CfgNode -- a config container (config/config.py:1);
  * omegaconf.DictConfig -- isinstance check only (utils/config.py:7,143);
  * cv2 / imageio / torchvision / equilib -- imported by utils/panocam.py:1-15
    and utils/utils.py:1, never called by inference().

`install()` registers in-memory stub modules and puts /root/reference on
sys.path.  It is only used by oracle/gen_golden.py and by the
reference-vs-oracle test that auto-skips when /root/reference is absent (the
GPU box).  torch.hub.load_state_dict_from_url is patched so the reference's
constructor (perspectivefields.py:178-192) loads a caller-provided checkpoint
instead of downloading.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PF_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "perspective2d"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs() -> None:
    import torch
    import yaml
    from torch import nn

    # ---- timm.models.layers -------------------------------------------------
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            assert not self.training, "shim DropPath is inference-only"
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    if "timm" not in sys.modules:
        timm = _mod("timm")
        models = _mod("timm.models")
        layers = _mod(
            "timm.models.layers",
            DropPath=DropPath,
            to_2tuple=to_2tuple,
            trunc_normal_=trunc_normal_,
        )
        timm.models = models
        models.layers = layers
        _mod("timm.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)

    # ---- yacs.config.CfgNode -------------------------------------------------
    class CfgNode(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            if self.__dict__.get("_frozen", False):
                raise AttributeError("CfgNode is frozen")
            self[k] = v

        def _merge(self, other: dict):
            for k, v in other.items():
                if k not in self:
                    raise KeyError(f"Non-existent config key: {k}")
                if isinstance(v, dict):
                    self[k]._merge(v)
                else:
                    self[k] = v

        def merge_from_file(self, path):
            with open(path) as f:
                self._merge(yaml.safe_load(f))

        def freeze(self):
            for v in self.values():
                if isinstance(v, CfgNode):
                    v.freeze()
            self.__dict__["_frozen"] = True

    if "yacs" not in sys.modules:
        yacs = _mod("yacs")
        yacs.config = _mod("yacs.config", CfgNode=CfgNode)

    if "omegaconf" not in sys.modules:
        _mod("omegaconf", DictConfig=type("DictConfig", (), {}))

    # ---- import-only modules ---------------------------------------------------
    for name in ("cv2", "imageio"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _mod(name)
    if "torchvision" not in sys.modules:
        try:
            __import__("torchvision")
        except Exception:
            tv = _mod("torchvision")
            tv.transforms = _mod("torchvision.transforms")
    if "equilib" not in sys.modules:
        _mod("equilib", __version__="0.3.0", equi2pers=None, grid_sample=None)


def install() -> None:
    """Make `import perspective2d` resolve to the unmodified reference."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    _install_stubs()
    # an alias package of the same name lives at the repo root; the reference
    # must win inside the process that called install()
    for k in [k for k in sys.modules if k == "perspective2d" or k.startswith("perspective2d.")]:
        del sys.modules[k]
    if REFERENCE_ROOT in sys.path:
        sys.path.remove(REFERENCE_ROOT)
    sys.path.insert(0, REFERENCE_ROOT)


def build_reference(version: str, state_dict: dict):
    """Instantiate the reference `PerspectiveFields(version)` on CPU with `state_dict`
    loaded the way its own constructor does (checkpoint = {"model": state_dict})."""
    install()
    import torch
    import torch.hub

    orig = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda *a, **k: {"model": state_dict}
    try:
        from perspective2d import PerspectiveFields  # the reference's class

        model = PerspectiveFields(version).eval()
    finally:
        torch.hub.load_state_dict_from_url = orig
    # strict check the reference itself skips (perspectivefields.py:185,192)
    missing = set(model.state_dict().keys()) - set(state_dict.keys())
    unexpected = set(state_dict.keys()) - set(model.state_dict().keys())
    if missing or unexpected:
        raise RuntimeError(f"synthetic checkpoint mismatch: missing={sorted(missing)[:5]} unexpected={sorted(unexpected)[:5]}")
    return model
