"""Checkpoint schema of the reference's `{"model": state_dict}` files.

The key names are the reference module tree's (perspectivefields.py:121-150,
mix_transformers.py:252-402, decode_head.py:42-54,224-288, gravity_head.py:65-117,
latitude_head.py:65-118, convnext.py:62-126, param_network.py:34-44,171-191).
The schema is *generated* from the architecture constants, then verified against
the reference's own `state_dict()` by tests/test_oracle_vs_reference.py whenever
/root/reference is importable.  The engine's loader is strict against it (the
reference loads with strict=False, perspectivefields.py:185,192).
"""
from __future__ import annotations

from collections import OrderedDict

from .config import arch_of, get_cfg

# MiT-B3 (mix_transformers.py:511-524)
MIT_DIMS = (64, 128, 320, 512)
MIT_HEADS = (1, 2, 5, 8)
MIT_DEPTHS = (3, 4, 18, 3)
MIT_SR = (8, 4, 2, 1)
MIT_PATCH = ((7, 4, 3), (3, 2, 1), (3, 2, 1), (3, 2, 1))  # (kernel, stride, pad) mix_transformers.py:274-302
MIT_MLP_RATIO = 4
# decoder (gravity_head.py:121-137)
DEC_EMBED = 768
DEC_FEAT = 256
LL_CH = 64
# ConvNeXt-T (convnext.py:78-79)
CNX_DEPTHS = (3, 3, 9, 3)
CNX_DIMS = (96, 192, 384, 768)


def _mit(sd, pfx="backbone."):
    cin = 3
    for s in range(4):
        C, (k, _, _) = MIT_DIMS[s], MIT_PATCH[s]
        p = f"{pfx}patch_embed{s + 1}."
        sd[p + "proj.weight"] = (C, cin, k, k)
        sd[p + "proj.bias"] = (C,)
        sd[p + "norm.weight"] = (C,)
        sd[p + "norm.bias"] = (C,)
        for i in range(MIT_DEPTHS[s]):
            b = f"{pfx}block{s + 1}.{i}."
            sd[b + "norm1.weight"] = (C,)
            sd[b + "norm1.bias"] = (C,)
            sd[b + "attn.q.weight"] = (C, C)
            sd[b + "attn.q.bias"] = (C,)
            sd[b + "attn.kv.weight"] = (2 * C, C)
            sd[b + "attn.kv.bias"] = (2 * C,)
            sd[b + "attn.proj.weight"] = (C, C)
            sd[b + "attn.proj.bias"] = (C,)
            if MIT_SR[s] > 1:
                r = MIT_SR[s]
                sd[b + "attn.sr.weight"] = (C, C, r, r)
                sd[b + "attn.sr.bias"] = (C,)
                sd[b + "attn.norm.weight"] = (C,)
                sd[b + "attn.norm.bias"] = (C,)
            sd[b + "norm2.weight"] = (C,)
            sd[b + "norm2.bias"] = (C,)
            Hd = MIT_MLP_RATIO * C
            sd[b + "mlp.fc1.weight"] = (Hd, C)
            sd[b + "mlp.fc1.bias"] = (Hd,)
            sd[b + "mlp.dwconv.dwconv.weight"] = (Hd, 1, 3, 3)
            sd[b + "mlp.dwconv.dwconv.bias"] = (Hd,)
            sd[b + "mlp.fc2.weight"] = (C, Hd)
            sd[b + "mlp.fc2.bias"] = (C,)
        sd[f"{pfx}norm{s + 1}.weight"] = (C,)
        sd[f"{pfx}norm{s + 1}.bias"] = (C,)
        cin = C


def _ll(sd):
    sd["ll_enc.conv1.weight"] = (LL_CH, 3, 7, 7)
    sd["ll_enc.bn1.weight"] = (LL_CH,)
    sd["ll_enc.bn1.bias"] = (LL_CH,)
    sd["ll_enc.bn1.running_mean"] = (LL_CH,)
    sd["ll_enc.bn1.running_var"] = (LL_CH,)
    sd["ll_enc.bn1.num_batches_tracked"] = ()


def _head(sd, name, n_out):
    p = f"persformer_heads.{name}_head."
    for k in (4, 3, 2, 1):  # construction order in gravity_head.py:65-68
        sd[p + f"linear_c{k}.proj.weight"] = (DEC_EMBED, MIT_DIMS[k - 1])
        sd[p + f"linear_c{k}.proj.bias"] = (DEC_EMBED,)
    for k in (4, 3, 2, 1):
        sd[p + f"linear_c{k}_proc.weight"] = (DEC_FEAT, DEC_EMBED, 3, 3)
        sd[p + f"linear_c{k}_proc.bias"] = (DEC_FEAT,)
    for k in (1, 2, 3, 4):
        units = ("resConfUnit2",) if k == 4 else ("resConfUnit1", "resConfUnit2")
        for u in units:
            for c in ("conv1", "conv2"):
                sd[p + f"fusion{k}.{u}.{c}.weight"] = (DEC_FEAT, DEC_FEAT, 3, 3)
                sd[p + f"fusion{k}.{u}.{c}.bias"] = (DEC_FEAT,)
    sd[p + "conv_fuse_conv0.conv.weight"] = (64, DEC_FEAT + LL_CH, 3, 3)
    sd[p + "conv_fuse_conv0.conv.bias"] = (64,)
    sd[p + "conv_fuse_conv1.conv.weight"] = (32, 64, 3, 3)
    sd[p + "conv_fuse_conv1.conv.bias"] = (32,)
    sd[p + f"linear_pred_{name}.weight"] = (n_out, 32, 1, 1)
    sd[p + f"linear_pred_{name}.bias"] = (n_out,)


def _convnext(sd, n_out, pfx="param_net.backbone."):
    d = CNX_DIMS
    sd[pfx + "downsample_layers.0.0.weight"] = (d[0], 3, 4, 4)
    sd[pfx + "downsample_layers.0.0.bias"] = (d[0],)
    sd[pfx + "downsample_layers.0.1.weight"] = (d[0],)
    sd[pfx + "downsample_layers.0.1.bias"] = (d[0],)
    for i in range(1, 4):
        sd[pfx + f"downsample_layers.{i}.0.weight"] = (d[i - 1],)
        sd[pfx + f"downsample_layers.{i}.0.bias"] = (d[i - 1],)
        sd[pfx + f"downsample_layers.{i}.1.weight"] = (d[i], d[i - 1], 2, 2)
        sd[pfx + f"downsample_layers.{i}.1.bias"] = (d[i],)
    for s in range(4):
        C = d[s]
        for j in range(CNX_DEPTHS[s]):
            b = pfx + f"stages.{s}.{j}."
            sd[b + "gamma"] = (C,)
            sd[b + "dwconv.weight"] = (C, 1, 7, 7)
            sd[b + "dwconv.bias"] = (C,)
            sd[b + "norm.weight"] = (C,)
            sd[b + "norm.bias"] = (C,)
            sd[b + "pwconv1.weight"] = (4 * C, C)
            sd[b + "pwconv1.bias"] = (4 * C,)
            sd[b + "pwconv2.weight"] = (C, 4 * C)
            sd[b + "pwconv2.bias"] = (C,)
    sd[pfx + "norm.weight"] = (d[3],)
    sd[pfx + "norm.bias"] = (d[3],)
    sd[pfx + "head.weight"] = (n_out, d[3])
    sd[pfx + "head.bias"] = (n_out,)


def schema_for_arch(arch: dict) -> "OrderedDict[str, tuple]":
    sd: "OrderedDict[str, tuple]" = OrderedDict()
    _mit(sd)
    _ll(sd)
    _head(sd, "gravity", arch["gravity_out"])
    _head(sd, "latitude", arch["latitude_out"])
    if arch["param_net"] is not None:
        _convnext(sd, arch["param_out"])
    return sd


def checkpoint_schema(version: str) -> "OrderedDict[str, tuple]":
    return schema_for_arch(arch_of(get_cfg(version)))


OPTIONAL_KEYS = ("ll_enc.bn1.num_batches_tracked",)


def validate_state_dict(version: str, state_dict, strict: bool = True):
    """Every schema key must be present with the right shape (always fatal: a partially initialised network is the
    silent failure mode of the reference's strict=False load, perspectivefields.py:185,192).  Keys the architecture
    does not have are an error when `strict`, otherwise they are returned so that the caller can drop them (real zoo
    files may carry trainer state or unused heads)."""
    sch = checkpoint_schema(version)
    missing = [k for k in sch if k not in state_dict and k not in OPTIONAL_KEYS]
    extra = [k for k in state_dict if k not in sch]
    bad = [
        (k, tuple(state_dict[k].shape), sch[k])
        for k in sch
        if k in state_dict and tuple(state_dict[k].shape) != tuple(sch[k])
    ]
    if missing or bad or (extra and strict):
        raise ValueError(
            f"checkpoint does not match schema for {version}: "
            f"missing={missing[:4]} ({len(missing)}), unexpected={extra[:4]} ({len(extra)}), "
            f"shape mismatches={bad[:4]} ({len(bad)})"
        )
    return extra
