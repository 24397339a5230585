"""Model zoo and configuration for the MI355X-native PerspectiveFields path.

Mirrors what the reference exposes through `PerspectiveFields.cfg` and
`model_zoo` (reference: perspective2d/perspectivefields.py:86-118,
perspective2d/config/config.py:4-137 and the five YAML overlays in
perspective2d/config/).  Only keys that influence inference are modelled; the
reference's training/FPN/visualisation keys are not part of the hot path.

The config object supports attribute access (`cfg.MODEL.GRAVITY_DECODER.LOSS_TYPE`)
like the reference's yacs CfgNode, and is frozen after construction.
"""
from __future__ import annotations

import copy

NET_H = 320  # the network always runs at 320x320 (every reference YAML: DATALOADER.RESIZE)
NET_W = 320


class CfgNode(dict):
    """Minimal attribute-access config node (yacs is not a dependency here)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        if self.__dict__.get("_frozen", False):
            raise AttributeError(f"config is frozen; cannot set {key}")
        self[key] = value

    def freeze(self):
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()
        self.__dict__["_frozen"] = True
        return self

    def is_frozen(self):
        return self.__dict__.get("_frozen", False)


def _node(d):
    if isinstance(d, dict):
        return CfgNode({k: _node(v) for k, v in d.items()})
    return copy.deepcopy(d)


_DEFAULTS = {
    "VIS_PERIOD": 100,
    "DEBUG_ON": False,
    "INPUT": {"FORMAT": "BGR", "ONLINE_CROP": False},
    "DATALOADER": {"RESIZE": [NET_H, NET_W]},
    "MODEL": {
        "WEIGHTS": "",
        "GRAVITY_ON": True,
        "LATITUDE_ON": True,
        "RECOVER_RPF": False,
        "RECOVER_PP": False,
        "FREEZE": [],
        "PIXEL_MEAN": [103.53, 116.28, 123.675],  # BGR
        "PIXEL_STD": [1.0, 1.0, 1.0],
        "BACKBONE": {"NAME": "mitb3"},
        "PERSFORMER_HEADS": {"NAME": "StandardPersformerHeads"},
        "GRAVITY_DECODER": {
            "NAME": "GravityDecoder",
            "LOSS_TYPE": "classification",
            "LOSS_WEIGHT": 1.0,
            "NUM_CLASSES": 73,
            "IGNORE_VALUE": 72,
        },
        "LATITUDE_DECODER": {
            "NAME": "LatitudeDecoder",
            "LOSS_TYPE": "regression",
            "LOSS_WEIGHT": 1.0,
            "NUM_CLASSES": 1,
            "IGNORE_VALUE": -1,
        },
        "PARAM_DECODER": {
            "NAME": "ParamNet",
            "LOSS_TYPE": "regression",
            "LOSS_WEIGHT": 1.0,
            "PREDICT_PARAMS": ["roll", "pitch", "rel_focal", "rel_cx", "rel_cy"],
            "INPUT_SIZE": 320,
        },
    },
}

_REG_HEADS = {
    "GRAVITY_DECODER": {"LOSS_TYPE": "regression", "NUM_CLASSES": 73},
    "LATITUDE_DECODER": {"LOSS_TYPE": "regression", "NUM_CLASSES": 1},
}

# overlays equivalent to the reference's YAML files (inference-relevant keys only)
_OVERLAYS = {
    # paramnet_360cities_edina_rpf.yaml / paramnet_gsv_rpf.yaml
    "rpf": {
        "MODEL": {
            **_REG_HEADS,
            "RECOVER_RPF": True,
            "RECOVER_PP": False,
            "PARAM_DECODER": {"NAME": "ParamNet", "PREDICT_PARAMS": ["roll", "pitch", "vfov"], "INPUT_SIZE": 64},
        }
    },
    # paramnet_360cities_edina_rpfpp.yaml / paramnet_gsv_rpfpp.yaml
    "rpfpp": {
        "MODEL": {
            **_REG_HEADS,
            "RECOVER_RPF": True,
            "RECOVER_PP": True,
            "PARAM_DECODER": {
                "NAME": "ParamNetConvNextRegress",
                "PREDICT_PARAMS": ["roll", "pitch", "general_vfov", "rel_cx", "rel_cy"],
                "INPUT_SIZE": 64,
            },
        }
    },
    # cvpr2023.yaml
    "cvpr2023": {
        "MODEL": {
            "GRAVITY_DECODER": {"LOSS_TYPE": "classification", "NUM_CLASSES": 73},
            "LATITUDE_DECODER": {"LOSS_TYPE": "classification", "NUM_CLASSES": 180},
            "RECOVER_RPF": False,
            "RECOVER_PP": False,
        }
    },
}

_HF = "https://huggingface.co/spaces/jinlinyi/PerspectiveFields/resolve/main/models/"

# same keys / fields as the reference's model_zoo (perspectivefields.py:86-118)
model_zoo = {
    "Paramnet-360Cities-edina-centered": {
        "weights": _HF + "paramnet_360cities_edina_rpf.pth",
        "config_file": "paramnet_360cities_edina_rpf.yaml",
        "overlay": "rpf",
        "param": True,
        "description": "Trained on 360cities and EDINA dataset. Assumes centered principal point. Predicts roll, pitch and fov.",
    },
    "Paramnet-360Cities-edina-uncentered": {
        "weights": _HF + "paramnet_360cities_edina_rpfpp.pth",
        "config_file": "paramnet_360cities_edina_rpfpp.yaml",
        "overlay": "rpfpp",
        "param": True,
        "description": "Trained on 360cities and EDINA dataset. Predicts roll, pitch, fov and principal point.",
    },
    "PersNet-360Cities": {
        "weights": _HF + "cvpr2023.pth",
        "config_file": "cvpr2023.yaml",
        "overlay": "cvpr2023",
        "param": False,
        "description": "Trained on 360cities. Predicts perspective fields.",
    },
    "PersNet_Paramnet-GSV-uncentered": {
        "weights": _HF + "paramnet_gsv_rpfpp.pth",
        "config_file": "paramnet_gsv_rpfpp.yaml",
        "overlay": "rpfpp",
        "param": True,
        "description": "Trained on GSV. Predicts roll, pitch, fov and principal point.",
    },
    "PersNet_Paramnet-GSV-centered": {
        "weights": _HF + "paramnet_gsv_rpf.pth",
        "config_file": "paramnet_gsv_rpf.yaml",
        "overlay": "rpf",
        "param": True,
        "description": "Trained on GSV. Assumes centered principal point. Predicts roll, pitch and fov.",
    },
}


def _merge(dst: CfgNode, src: dict):
    for k, v in src.items():
        if k not in dst:
            raise KeyError(f"Non-existent config key: {k}")
        if isinstance(v, dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def get_cfg_defaults() -> CfgNode:
    return _node(_DEFAULTS)


def get_cfg(version: str) -> CfgNode:
    """Frozen config for a zoo version.  Unknown version -> KeyError, as in the
    reference (perspectivefields.py:127 indexes model_zoo[version])."""
    entry = model_zoo[version]
    cfg = get_cfg_defaults()
    _merge(cfg, _OVERLAYS[entry["overlay"]])
    return cfg.freeze()


# ---- architecture descriptor used by schema / engine / oracle ----------------
ARCH_PARAMNET_CENTERED = 0   # regression heads + ParamNet (5 raw outputs, uses 0..2)
ARCH_PERSNET_CLS = 1         # classification heads (73 / 180 logits), no ParamNet
ARCH_PARAMNET_UNCENTERED = 2  # regression heads + ParamNetConvNextRegress (64x64 input)


def arch_of(cfg: CfgNode) -> dict:
    """Derive the tensor-level architecture from a config (what the reference's
    builders decide in gravity_head.py:41-117, latitude_head.py:41-118,
    param_network.py:11-19,34-44,171-191)."""
    g, l = cfg.MODEL.GRAVITY_DECODER, cfg.MODEL.LATITUDE_DECODER
    g_cls = g.LOSS_TYPE == "classification"
    l_cls = l.LOSS_TYPE == "classification"
    # gravity_head.py:62-63: regression forces 2 channels; latitude_head.py:52-53 is a
    # no-op comparison, so the latitude channel count is NUM_CLASSES from the config.
    g_out = g.NUM_CLASSES if g_cls else 2
    l_out = l.NUM_CLASSES
    param = bool(cfg.MODEL.RECOVER_RPF or cfg.MODEL.RECOVER_PP)
    pname = cfg.MODEL.PARAM_DECODER.NAME if param else None
    if pname is None:
        arch_id, p_out = ARCH_PERSNET_CLS, 0
    elif pname == "ParamNet":
        arch_id, p_out = ARCH_PARAMNET_CENTERED, 5
    elif pname == "ParamNetConvNextRegress":
        arch_id, p_out = ARCH_PARAMNET_UNCENTERED, len(cfg.MODEL.PARAM_DECODER.PREDICT_PARAMS)
    else:
        raise ValueError(f"Unknown paramnet name: {pname}")
    if g_cls != l_cls:
        raise ValueError("mixed regression/classification heads are not a zoo configuration")
    return {
        "arch_id": arch_id,
        "gravity_cls": g_cls,
        "latitude_cls": l_cls,
        "gravity_out": g_out,
        "latitude_out": l_out,
        "param_net": pname,
        "param_out": p_out,
        "param_input_size": cfg.MODEL.PARAM_DECODER.INPUT_SIZE if pname == "ParamNetConvNextRegress" else NET_H,
        "predict_params": list(cfg.MODEL.PARAM_DECODER.PREDICT_PARAMS) if param else [],
    }
