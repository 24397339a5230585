"""Model zoo and configuration for the MI355X-native PerspectiveFields path.

Mirrors what the reference exposes through `PerspectiveFields.cfg` and
`model_zoo` (reference: perspective2d/perspectivefields.py:86-118,
perspective2d/config/config.py:4-137 and the five YAML overlays in
perspective2d/config/).  Only keys that influence inference are modelled; the
reference's training/FPN/visualisation keys are not part of the hot path.
YAML files in the reference's format are read (`CfgNode.merge_from_file`,
`get_cfg_from_file`) and written (`CfgNode.dump`, `write_zoo_yamls`): the
reference's five files load to exactly the zoo configs below.

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

    # ---- YAML in the reference's format (perspective2d/config/*.yaml are yacs dumps: nested mappings, keys sorted) ----
    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else copy.deepcopy(v)) for k, v in self.items()}

    def dump(self) -> str:
        """YAML text of this config, keys sorted at every level -- what yacs' `cfg.dump()` writes and what `merge_from_file` reads back."""
        import yaml

        return yaml.safe_dump(self.to_dict(), default_flow_style=False, sort_keys=True)

    def merge_from_file(self, path: str):
        """Overlay a YAML file in the reference's format (the way the reference builds its config: defaults, then `cfg.merge_from_file(<zoo yaml>)`,
        perspectivefields.py:129-132).  Keys of the modelled sections must exist, as with yacs ("Non-existent config key"); the sections of a reference YAML that
        only matter to training / evaluation (TRAINING_ONLY_SECTIONS) are not part of this config and are skipped -- their names are returned."""
        import yaml

        if self.is_frozen():
            raise AttributeError("config is frozen; cannot merge")
        with open(path) as f:
            src = yaml.safe_load(f) or {}
        if not isinstance(src, dict):
            raise ValueError(f"{path}: a config file is a YAML mapping")
        skipped = sorted(k for k in src if k in TRAINING_ONLY_SECTIONS)
        _merge(self, {k: v for k, v in src.items() if k not in TRAINING_ONLY_SECTIONS})
        return skipped


# top-level sections of the reference's YAML files / detectron2-style defaults that do not influence inference (data sets, optimiser, evaluation schedule)
TRAINING_ONLY_SECTIONS = ("DATASETS", "SOLVER", "TEST", "SEED", "OUTPUT_DIR", "CUDNN_BENCHMARK", "VERSION", "GLOBAL")


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
    # paramnet_gsv_rpf.yaml: as "rpf", but PARAM_DECODER.INPUT_SIZE stays 320 in that file (ParamNet does not read it: its input is the 320x320 field map)
    "rpf_gsv": {
        "MODEL": {
            **_REG_HEADS,
            "RECOVER_RPF": True,
            "RECOVER_PP": False,
            "PARAM_DECODER": {"NAME": "ParamNet", "PREDICT_PARAMS": ["roll", "pitch", "vfov"], "INPUT_SIZE": 320},
        }
    },
    # paramnet_gsv_rpfpp.yaml: as "rpfpp" with PARAM_DECODER.LOSS_WEIGHT 0.1 (a training value; kept so that the config equals the file's)
    "rpfpp_gsv": {
        "MODEL": {
            **_REG_HEADS,
            "RECOVER_RPF": True,
            "RECOVER_PP": True,
            "PARAM_DECODER": {
                "NAME": "ParamNetConvNextRegress",
                "PREDICT_PARAMS": ["roll", "pitch", "general_vfov", "rel_cx", "rel_cy"],
                "INPUT_SIZE": 64,
                "LOSS_WEIGHT": 0.1,
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
        "overlay": "rpfpp_gsv",
        "param": True,
        "description": "Trained on GSV. Predicts roll, pitch, fov and principal point.",
    },
    "PersNet_Paramnet-GSV-centered": {
        "weights": _HF + "paramnet_gsv_rpf.pth",
        "config_file": "paramnet_gsv_rpf.yaml",
        "overlay": "rpf_gsv",
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


def get_cfg_from_file(path: str) -> CfgNode:
    """Frozen config from a YAML file in the reference's format (defaults + the file, like the reference's constructor)."""
    cfg = get_cfg_defaults()
    cfg.merge_from_file(path)
    return cfg.freeze()


def write_zoo_yamls(directory: str):
    """Write the zoo configs as YAML files in the reference's format and under the reference's file names (one per distinct config_file) into `directory`;
    returns {file name: path}.  The files are generated from the overlays above, not shipped: `get_cfg_from_file` reads them back to exactly `get_cfg(version)`,
    and it reads the reference's own perspective2d/config/*.yaml to the same configs (tests/test_host_logic.py)."""
    import os

    os.makedirs(directory, exist_ok=True)
    done = {}
    for version, entry in model_zoo.items():
        if entry["config_file"] in done:
            continue
        cfg = get_cfg_defaults()
        _merge(cfg, _OVERLAYS[entry["overlay"]])
        path = os.path.join(directory, entry["config_file"])
        with open(path, "w") as f:
            f.write(cfg.dump())
        done[entry["config_file"]] = path
    return done


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
