"""Deterministic synthetic checkpoints (there is no network: the reference's trained
weights, perspectivefields.py:86-118, cannot be downloaded).

The same seeded tensors are loaded into (a) the unmodified reference when golden
vectors are generated, (b) the oracle and (c) the HIP engine, through the schema in
schema.py.  The generator is numpy-only (PCG64 streams are platform independent) and
draws one independent stream per key, so adding keys never perturbs other tensors.

Initialisation is variance preserving (gain/sqrt(fan_in)) rather than the
reference's training init (trunc-normal 0.02, gamma=1e-6, convnext.py:28), so that
every residual branch, LayerNorm affine, BatchNorm statistic and layer-scale gamma
contributes measurably to the outputs -- a bug in any of them shows up in parity
tests instead of hiding below rounding error.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .schema import checkpoint_schema


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def _normal(rng, shape, std):
    return (rng.standard_normal(shape, dtype=np.float64) * std).astype(np.float32)


def synthetic_state_dict(version: str, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    sch = checkpoint_schema(version)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in sch.items():
        rng = _rng(seed, key)
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("num_batches_tracked"):
            out[key] = np.asarray(1000, dtype=np.int64)
            continue
        if key.endswith("running_mean"):
            out[key] = _normal(rng, shape, 0.2)
            continue
        if key.endswith("running_var"):
            out[key] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            continue
        if leaf == "gamma":  # ConvNeXt layer scale (convnext.py:39-43)
            out[key] = rng.uniform(0.05, 0.35, shape).astype(np.float32)
            continue
        # affine of a LayerNorm / BatchNorm: the sibling ".weight" is 1-D
        is_norm = leaf in ("weight", "bias") and len(sch.get(key.rsplit(".", 1)[0] + ".weight", (0, 0))) == 1
        if is_norm:
            if leaf == "weight":
                out[key] = (1.0 + _normal(rng, shape, 0.1)).astype(np.float32)
            else:
                out[key] = _normal(rng, shape, 0.1)
            continue
        if leaf == "bias":
            if key.endswith("param_net.backbone.head.bias"):
                b = _normal(rng, shape, 0.05)
                if shape[0] >= 3:
                    b[2] += 0.6  # keeps 1/(2 tan(x2)) (param_network.py:66) well conditioned
                out[key] = b
            else:
                out[key] = _normal(rng, shape, 0.05)
            continue
        # weights
        fan_in = int(np.prod(shape[1:]))
        gain = 1.0
        if "dwconv" in key:
            fan_in = int(np.prod(shape[2:]))
        if (
            key.endswith(("fc1.weight", "pwconv1.weight"))
            or "conv_fuse" in key
            or ("resConfUnit" in key and ".conv1." in key)
        ):
            gain = 1.4  # followed by ReLU / GELU
        if key.endswith(("attn.proj.weight", "fc2.weight", "pwconv2.weight")):
            gain = 0.5  # residual-branch outputs
        if key in ("backbone.patch_embed1.proj.weight", "ll_enc.conv1.weight"):
            gain = 1.0 / 64.0  # inputs are mean-subtracted 0..255 pixels
        if "linear_pred_gravity" in key:
            gain = 1.0
        if "linear_pred_latitude" in key:
            gain = 0.2
        if key.endswith("param_net.backbone.head.weight"):
            gain = 0.15
        out[key] = _normal(rng, shape, gain / np.sqrt(fan_in))
    return out


def heavy_tailed_state_dict(version: str, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """The synthetic checkpoint made to LOOK like a trained one where it matters for the split-f16 contractions (trunc-normal weights never produce what
    trained transformers do): per-output-channel weight magnitudes spread over two orders of magnitude (log-normal, sigma 1), 0.5 % of the entries of every dense
    weight 8x larger (heavy tail), and OUTLIER CHANNELS in the token streams -- two channels per MiT block get a +-25 bias on `attn.proj` / `mlp.fc2`, the same two in every block of a stage (they pile up
    along the residual stream to tens of sigma: what the fused LayerNorms then see as raw rows), two LayerNorm gains per norm layer are 20x.  Same schema, same generator streams."""
    sd = synthetic_state_dict(version, seed)
    for key in list(sd):
        v = sd[key]
        rng = _rng(seed + 7919, key)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "weight" and v.ndim >= 2 and "dwconv" not in key and "linear_pred" not in key and not key.endswith("head.weight"):
            ch = np.exp(rng.standard_normal(v.shape[0]) * 1.0).astype(np.float32)
            ch /= np.sqrt(np.mean(ch.astype(np.float64) ** 2)).astype(np.float32)  # same overall gain
            w = v * ch.reshape((-1,) + (1,) * (v.ndim - 1))
            big = rng.random(v.shape) < 0.005
            sd[key] = np.where(big, w * np.float32(8.0), w).astype(np.float32)
        elif leaf == "bias" and key.startswith("backbone.block") and key.endswith(("attn.proj.bias", "mlp.fc2.bias")):
            b = v.copy()
            srng = _rng(seed + 7919, key.split(".")[1])  # the SAME two channels (and signs) in every block of a stage: they pile up along the residual stream
            idx = srng.choice(b.shape[0], 2, replace=False)
            b[idx] += np.float32(25.0) * np.where(srng.random(2) < 0.5, -1.0, 1.0).astype(np.float32)
            sd[key] = b
        elif leaf == "weight" and v.ndim == 1 and ("norm" in key) and "bn1" not in key:
            g = v.copy()
            idx = rng.choice(g.shape[0], 2, replace=False)
            g[idx] *= np.float32(20.0)
            sd[key] = g
    return sd


def to_torch(state_dict):
    import torch

    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in state_dict.items())


def synthetic_image(h: int, w: int, seed: int = 0, smooth: bool = True) -> np.ndarray:
    """Synthetic HxWx3 uint8 BGR image: a sum of random low-frequency
    sinusoids plus noise (smooth=True), or i.i.d. uniform bytes."""
    rng = np.random.Generator(np.random.PCG64([seed, h, w, int(smooth)]))
    if not smooth:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float64), np.linspace(0, 1, w, dtype=np.float64), indexing="ij")
    img = np.zeros((h, w, 3), dtype=np.float64)
    for c in range(3):
        for _ in range(8):
            fx, fy = rng.uniform(-6, 6, 2)
            ph = rng.uniform(0, 2 * np.pi)
            img[:, :, c] += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)
    img = (img - img.min()) / (img.max() - img.min())
    img = img * 0.9 + 0.05 + rng.normal(0, 0.05, img.shape)
    return np.clip(np.rint(img * 255), 0, 255).astype(np.uint8)
