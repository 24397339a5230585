"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the PerspectiveFields inference path.

A functional restatement (torch CPU ops on a plain state_dict; no nn.Module tree) of
the reference's algorithm for the path named by BASELINE.json: MiT-B3 backbone ->
gravity/latitude decoders -> ParamNet -> post-process.  Every function cites the
reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product path (perspectivefields_amd/)
never does and fails loudly without its HIP extension.

Pinning: this oracle is checked against (a) golden vectors produced by the unmodified
reference in this container (oracle/gen_golden.py -> tests/golden/*.npz,
tests/test_oracle_golden.py) and (b) the live reference when /root/reference is
importable (tests/test_oracle_vs_reference.py).  The reference itself has no tests or
golden vectors for this path; its two printed known-answer triples
need trained weights that cannot be downloaded here.

All activations are NCHW / (B,N,C) torch tensors as in the reference; `dtype` may be
torch.float32 (the reference's arithmetic) or torch.float64 (error-free yardstick).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

NET = 320
MIT_DIMS = (64, 128, 320, 512)
MIT_HEADS = (1, 2, 5, 8)
MIT_DEPTHS = (3, 4, 18, 3)
MIT_SR = (8, 4, 2, 1)
MIT_PATCH = ((7, 4, 3), (3, 2, 1), (3, 2, 1), (3, 2, 1))
CNX_DEPTHS = (3, 3, 9, 3)
CNX_DIMS = (96, 192, 384, 768)
PIXEL_MEAN_BGR = (103.53, 116.28, 123.675)  # config/config.py:77


class Weights:
    """state_dict view with a key prefix and dtype cast."""

    def __init__(self, sd, prefix="", dtype=torch.float32):
        self.sd, self.prefix, self.dtype = sd, prefix, dtype

    def sub(self, p):
        return Weights(self.sd, self.prefix + p, self.dtype)

    def __call__(self, name):
        t = self.sd[self.prefix + name]
        if not torch.is_tensor(t):
            t = torch.from_numpy(np.ascontiguousarray(t))
        return t.to(self.dtype)

    def has(self, name):
        return (self.prefix + name) in self.sd


# --------------------------------------------------------------------------- ops
def layer_norm(x, w, b, eps):
    """LayerNorm over the last dim, biased variance (torch.nn.LayerNorm; eps per
    SURVEY appendix C.3)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu(x):
    """Exact erf GELU (nn.GELU() default: mix_transformers.py:20, convnext.py:37)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def bilinear(x, size=None, scale=None):
    """align_corners=False bilinear, no antialias (decode_head.py:284, utils.py:504)."""
    return F.interpolate(x, size=size, scale_factor=scale, mode="bilinear", align_corners=False)


# ----------------------------------------------------------------------- MiT-B3
def mit_attention(w, x, H, W, heads, sr):
    """Spatial-reduction attention, mix_transformers.py:108-141."""
    B, N, C = x.shape
    d = C // heads
    q = F.linear(x, w("q.weight"), w("q.bias")).reshape(B, N, heads, d).transpose(1, 2)
    if sr > 1:
        xm = x.transpose(1, 2).reshape(B, C, H, W)
        xr = F.conv2d(xm, w("sr.weight"), w("sr.bias"), stride=sr)  # :88,118
        xr = xr.reshape(B, C, -1).transpose(1, 2)
        xr = layer_norm(xr, w("norm.weight"), w("norm.bias"), 1e-5)  # nn.LayerNorm default eps :89
    else:
        xr = x
    kv = F.linear(xr, w("kv.weight"), w("kv.bias")).reshape(B, -1, 2, heads, d)
    k = kv[:, :, 0].transpose(1, 2)  # (B, heads, M, d)
    v = kv[:, :, 1].transpose(1, 2)
    a = (q @ k.transpose(-2, -1)) * (d ** -0.5)  # :133
    a = a.softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, w("proj.weight"), w("proj.bias"))


def mit_mlp(w, x, H, W):
    """fc1 -> depthwise 3x3 -> GELU -> fc2, mix_transformers.py:49-56,502-508."""
    B, N, C = x.shape
    h = F.linear(x, w("fc1.weight"), w("fc1.bias"))
    Hd = h.shape[-1]
    hm = h.transpose(1, 2).reshape(B, Hd, H, W)
    hm = F.conv2d(hm, w("dwconv.dwconv.weight"), w("dwconv.dwconv.bias"), padding=1, groups=Hd)
    h = gelu(hm.flatten(2).transpose(1, 2))
    return F.linear(h, w("fc2.weight"), w("fc2.bias"))


def mit_block(w, x, H, W, heads, sr):
    """Pre-LN residual block (eps 1e-6 via mit_b3, mix_transformers.py:198-202,519)."""
    x = x + mit_attention(w.sub("attn."), layer_norm(x, w("norm1.weight"), w("norm1.bias"), 1e-6), H, W, heads, sr)
    x = x + mit_mlp(w.sub("mlp."), layer_norm(x, w("norm2.weight"), w("norm2.bias"), 1e-6), H, W)
    return x


def mit_b3(w, x, taps=None):
    """forward_features, mix_transformers.py:449-485.  Returns 4 NCHW maps.  taps: optional dict that receives every block output "mit.s<stage>.b<block>" as
    (B, H, W, C) -- the names and layout of the engine's shadow taps (include/pf_hip.h pf_debug_forward_u8)."""
    outs = []
    B = x.shape[0]
    for s in range(4):
        k, st, pd = MIT_PATCH[s]
        pe = w.sub(f"patch_embed{s + 1}.")
        x = F.conv2d(x, pe("proj.weight"), pe("proj.bias"), stride=st, padding=pd)  # :243
        H, W = x.shape[2:]
        x = x.flatten(2).transpose(1, 2)
        x = layer_norm(x, pe("norm.weight"), pe("norm.bias"), 1e-5)  # OverlapPatchEmbed.norm :224
        for i in range(MIT_DEPTHS[s]):
            x = mit_block(w.sub(f"block{s + 1}.{i}."), x, H, W, MIT_HEADS[s], MIT_SR[s])
            if taps is not None:
                taps[f"mit.s{s + 1}.b{i}"] = x.reshape(B, H, W, -1)
        x = layer_norm(x, w(f"norm{s + 1}.weight"), w(f"norm{s + 1}.bias"), 1e-6)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


def low_level_encoder(w, x):
    """conv7x7 s2 (no bias) -> BatchNorm(eval) -> ReLU, perspectivefields.py:70-83."""
    y = F.conv2d(x, w("conv1.weight"), None, stride=2, padding=3)
    y = F.batch_norm(y, w("bn1.running_mean"), w("bn1.running_var"), w("bn1.weight"), w("bn1.bias"), False, 0.0, 1e-5)
    return F.relu(y)


# ------------------------------------------------------------------- decoder heads
def residual_conv_unit(w, x):
    """decode_head.py:242-256.  The reference's ReLU is in-place, so the skip term
    is relu(x), not x: conv2(relu(conv1(relu(x)))) + relu(x)."""
    r = F.relu(x)
    t = F.relu(F.conv2d(r, w("conv1.weight"), w("conv1.bias"), padding=1))
    return F.conv2d(t, w("conv2.weight"), w("conv2.bias"), padding=1) + r


def feature_fusion(w, top, skip=None):
    """decode_head.py:271-288: out = top (+ RCU1(skip)); RCU2; bilinear x2."""
    o = top
    if skip is not None:
        o = o + residual_conv_unit(w.sub("resConfUnit1."), skip)
    o = residual_conv_unit(w.sub("resConfUnit2."), o)
    return bilinear(o, scale=2)


def decoder_layers(w, feats, ll, pred_name, taps=None):
    """GravityDecoder.layers / LatitudeDecoder.layers (gravity_head.py:139-176,
    latitude_head.py:138-175).  `feats` = [c1..c4] NCHW, `ll` = (B,64,160,160)."""
    fused = None
    for k in (4, 3, 2, 1):
        c = feats[k - 1]
        n, _, h, ww = c.shape
        e = F.linear(c.flatten(2).transpose(1, 2), w(f"linear_c{k}.proj.weight"), w(f"linear_c{k}.proj.bias"))
        e = e.transpose(1, 2).reshape(n, -1, h, ww)  # decode_head.py:51-53
        e = F.conv2d(e, w(f"linear_c{k}_proc.weight"), w(f"linear_c{k}_proc.bias"), padding=1)
        fused = feature_fusion(w.sub(f"fusion{k}."), e) if fused is None else feature_fusion(w.sub(f"fusion{k}."), fused, e)
    x = torch.cat([fused, ll], dim=1)  # gravity_head.py:170
    x = F.relu(F.conv2d(x, w("conv_fuse_conv0.conv.weight"), w("conv_fuse_conv0.conv.bias"), padding=1))
    if taps is not None:
        taps[f"dec.{pred_name}.conv0"] = x.permute(0, 2, 3, 1)
    x = bilinear(x, scale=2)
    x = F.relu(F.conv2d(x, w("conv_fuse_conv1.conv.weight"), w("conv_fuse_conv1.conv.bias"), padding=1))
    return F.conv2d(x, w(f"linear_pred_{pred_name}.weight"), w(f"linear_pred_{pred_name}.bias"))


def gravity_inference(w, feats, ll, classification, taps=None):
    """gravity_head.py:190-197 (the scale_factor=1 interpolate is an identity)."""
    x = decoder_layers(w, feats, ll, "gravity", taps)
    return x if classification else F.normalize(x, dim=1)


def latitude_inference(w, feats, ll, classification, taps=None):
    """latitude_head.py:189-193: regression output is sin(latitude) clamped to [-1,1]."""
    x = decoder_layers(w, feats, ll, "latitude", taps)
    return x if classification else torch.clamp(x, -1, 1)


def decode_gravity_bins(bins, num_bin):
    """utils/utils.py:114-130: bin -> (cos, sin) of bin*360/(num_bin-1) - 180 deg;
    bin num_bin-1 is 'invalid' -> (0,0)."""
    ang = (bins.to(torch.float64) * (360.0 / (num_bin - 1)) - 180.0) / 180.0 * math.pi
    vec = torch.stack((torch.cos(ang), torch.sin(ang)), dim=0).to(torch.float32)
    vec[:, bins == num_bin - 1] = 0
    return vec


def decode_latitude_bins(bins, num_classes):
    """utils/utils.py:148-162: bin centre -90 + (k + 0.5) * 180/num_classes degrees."""
    size = 180.0 / num_classes
    centers = torch.arange(-90, 90, size) + size / 2
    return centers[bins]


def postprocess_gravity(pred, height, width, classification, num_classes=73):
    """gravity_head.py:237-261 + utils.py:483-507 for one image: (decode) -> scale by
    (W/320, H/320) -> bilinear to (H,W) -> L2 normalise over the 2 channels."""
    vec = decode_gravity_bins(pred.argmax(dim=0), num_classes).to(pred.dtype) if classification else pred
    scale = torch.tensor([[width / NET], [height / NET]]).unsqueeze(-1).to(vec.dtype)  # float32 tensor in the reference
    v = vec * scale
    v = bilinear(v[None, :, :NET, :NET], size=(height, width))[0]
    return F.normalize(v, dim=0)


def postprocess_latitude(pred, height, width, classification, num_classes=180):
    """latitude_head.py:195-219 for one image -> (H,W) degrees."""
    if classification:
        lat = decode_latitude_bins(pred.argmax(dim=0), num_classes).to(pred.dtype)[None]
        return bilinear(lat[None, :, :NET, :NET], size=(height, width))[0, 0]
    lat = bilinear(pred[None, :, :NET, :NET], size=(height, width))[0, 0]
    return torch.rad2deg(torch.asin(lat))


# ------------------------------------------------------------------------ ConvNeXt
def convnext_ln_cf(x, wt, b, eps=1e-6):
    """channels_first LayerNorm, convnext.py:176-182."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return wt[:, None, None] * x + b[:, None, None]


def convnext_block(w, x):
    """convnext.py:46-59: dw7x7 -> LN -> Linear 4x -> GELU -> Linear -> gamma -> +res."""
    y = F.conv2d(x, w("dwconv.weight"), w("dwconv.bias"), padding=3, groups=x.shape[1])
    y = y.permute(0, 2, 3, 1)
    y = layer_norm(y, w("norm.weight"), w("norm.bias"), 1e-6)
    y = F.linear(y, w("pwconv1.weight"), w("pwconv1.bias"))
    y = gelu(y)
    y = F.linear(y, w("pwconv2.weight"), w("pwconv2.bias"))
    y = w("gamma") * y
    return x + y.permute(0, 3, 1, 2)


def convnext_tiny(w, x, taps=None):
    """ConvNeXt.forward, convnext.py:140-152 (depths 3,3,9,3; dims 96..768).  taps: optional dict receiving "pn.in" (padded to 4 channels like the engine's
    NHWC4 input), "pn.stem" and every block output "pn.s<stage>.b<block>" as (B, H, W, C)."""
    if taps is not None:
        taps["pn.in"] = F.pad(x.permute(0, 2, 3, 1), (0, 1))
    for s in range(4):
        ds = w.sub(f"downsample_layers.{s}.")
        if s == 0:
            x = F.conv2d(x, ds("0.weight"), ds("0.bias"), stride=4)
            x = convnext_ln_cf(x, ds("1.weight"), ds("1.bias"))
            if taps is not None:
                taps["pn.stem"] = x.permute(0, 2, 3, 1)
        else:
            x = convnext_ln_cf(x, ds("0.weight"), ds("0.bias"))
            x = F.conv2d(x, ds("1.weight"), ds("1.bias"), stride=2)
        for j in range(CNX_DEPTHS[s]):
            x = convnext_block(w.sub(f"stages.{s}.{j}."), x)
            if taps is not None:
                taps[f"pn.s{s + 1}.b{j}"] = x.permute(0, 2, 3, 1)
    x = layer_norm(x.mean([-2, -1]), w("norm.weight"), w("norm.bias"), 1e-6)
    return F.linear(x, w("head.weight"), w("head.bias"))


def general_vfov_to_focal(rel_cx, rel_cy, gvfov_deg):
    """utils/utils.py:47-91 with h=1, degree=True: solve cos_FoV(focal) = cos(gvfov)
    with scipy.fsolve from 1.5 (vector form), |focal| returned."""
    import scipy.optimize

    rel_cx = np.asarray(rel_cx, dtype=np.float64)
    rel_cy = np.asarray(rel_cy, dtype=np.float64)
    target = np.cos(np.radians(np.asarray(gvfov_deg, dtype=np.float64)))

    def fun(f):
        p = f ** 2 + rel_cx ** 2 + (rel_cy + 0.5) ** 2
        q = f ** 2 + rel_cx ** 2 + (rel_cy - 0.5) ** 2
        return (p + q - 1) / 2 / np.sqrt(p) / np.sqrt(q) - target

    return np.abs(scipy.optimize.fsolve(fun, np.ones(len(rel_cx)) * 1.5))


def fields_from_params(roll, pitch, vfov, rel_cx, rel_cy, im_h, im_w, mode="deg"):
    """Camera parameters -> (up field (im_h, im_w, 2), latitude map (im_h, im_w) in degrees): the step the reference's
    demos run right after inference (utils/utils.py:325-381 draw_from_r_p_f_cx_cy -> general_vfov_to_focal(:47-91),
    PanoCam.get_up_general (utils/panocam.py:451-512), PanoCam.get_lat_general (:514-556)), restated in float64 numpy.
    Quirks kept: the up field is sampled at pixel centres (j + 0.5), the latitude map on np.linspace(-c, size - c, size)
    (end points included, spacing size / (size - 1)); elevation == 0 gives a constant up field."""
    if mode == "deg":
        roll, pitch, vfov = np.radians(roll), np.radians(pitch), np.radians(vfov)
    elif mode != "rad":
        raise ValueError("Bad argument")
    # general_vfov_to_focal(rel_cx, rel_cy, h=1, gvfov, degree=False): same equation as general_vfov_to_focal above
    focal_rel = float(general_vfov_to_focal([rel_cx], [rel_cy], [np.degrees(vfov)])[0])
    elevation = pitch
    cx, cy = (rel_cx + 0.5) * im_w, (rel_cy + 0.5) * im_h
    focal_length = focal_rel * im_h
    # ---- get_up_general
    X = (np.linspace((-0.5 * im_w) + 0.5, (0.5 * im_w) - 0.5, im_w).reshape(1, im_w).repeat(im_h, 0).astype(np.float32) + 0.5 * im_w)
    Y = (np.linspace((-0.5 * im_h) + 0.5, (0.5 * im_h) - 0.5, im_h).reshape(im_h, 1).repeat(im_w, 1).astype(np.float32) + 0.5 * im_h)
    xy_cam = np.stack([X, Y], axis=2)
    if elevation == 0:
        up = np.ones(xy_cam.shape) * np.array([[-np.sin(roll)], [-np.cos(roll)]]).reshape((1, 2))
    else:
        vvp = np.array([[(np.sin(roll) * np.cos(elevation) * focal_length) / -np.sin(elevation) + cx],
                        [(np.cos(roll) * np.cos(elevation) * focal_length) / -np.sin(elevation) + cy]]).reshape((1, 2))
        up = (vvp - xy_cam) * np.sign(elevation)
    up = up / np.linalg.norm(up, axis=2)[:, :, None]
    # ---- get_lat_general
    dy = np.linspace((-im_h / 2) - (cy - (im_h / 2)), (im_h / 2) - (cy - (im_h / 2)), im_h)
    dx = np.linspace((-im_w / 2) - (cx - (im_w / 2)), (im_w / 2) - (cx - (im_w / 2)), im_w)
    x, y = np.meshgrid(dx, dy)
    x, y = x.ravel() / focal_length, y.ravel() / focal_length
    x_world = x * np.cos(roll) - y * np.sin(roll)
    y_world = x * np.cos(elevation) * np.sin(roll) + y * np.cos(elevation) * np.cos(roll) - np.sin(elevation)
    z_world = x * np.sin(elevation) * np.sin(roll) + y * np.sin(elevation) * np.cos(roll) + np.cos(elevation)
    lat = -np.arctan2(y_world, np.sqrt(x_world ** 2 + z_world ** 2)) / np.pi * 180
    return up, lat.reshape(im_h, im_w), focal_rel


def param_net(w, pred_gravity, pred_latitude, arch, taps=None):
    """ParamNet.forward (param_network.py:46-69) and ParamNetConvNextRegress.forward
    (param_network.py:193-221).  Input is the *normalised 320x320* up-vector and the
    clamped sin-latitude, not the post-processed fields."""
    x = torch.cat((pred_gravity, pred_latitude), dim=1)
    if arch["param_net"] == "ParamNet":
        y = convnext_tiny(w.sub("backbone."), x, taps)
        return {
            "raw": y,
            "pred_roll": y[:, 0] * 90.0,
            "pred_pitch": y[:, 1] * 90.0,
            "pred_vfov": y[:, 2] * 90.0,
            "pred_rel_focal": 1 / 2 / torch.tan(y[:, 2]),
        }
    size = arch["param_input_size"]
    x = F.interpolate(x, (size, size))  # nearest, param_network.py:197
    y = convnext_tiny(w.sub("backbone."), x, taps)
    factors = {"roll": 90.0, "pitch": 90.0, "vfov": 90.0, "rel_focal": 1.0, "rel_cx": 1.0, "rel_cy": 1.0, "general_vfov": 90.0}
    out = {"raw": y}
    for i, key in enumerate(arch["predict_params"]):
        out["pred_" + key] = y[:, i] * factors[key]
    if "pred_rel_focal" not in out:
        f = general_vfov_to_focal(out["pred_rel_cx"].double().numpy(), out["pred_rel_cy"].double().numpy(), out["pred_general_vfov"].double().numpy())
        out["pred_rel_focal"] = torch.tensor(f, dtype=torch.float32)
    return out


# -------------------------------------------------------------------------- driver
def normalise_input(images_bgr_f32, dtype=torch.float32):
    """(x - mean) / std, std = 1 (perspectivefields.py:235; config.py:77-78)."""
    mean = torch.tensor(PIXEL_MEAN_BGR, dtype=torch.float32).view(1, 3, 1, 1)
    return (images_bgr_f32.to(torch.float32) - mean).to(dtype)


def resize_to_net(img_bgr_u8):
    """ResizeTransform.apply_image (perspectivefields.py:34-46): PIL antialiased
    BILINEAR on uint8 to 320x320, aspect ratio not preserved."""
    from PIL import Image

    return np.asarray(Image.fromarray(img_bgr_u8).resize((NET, NET), Image.BILINEAR))


def forward(sd, arch, images_u8_320, sizes, dtype=torch.float32, stages=False, taps=None):
    """PerspectiveFields.forward (perspectivefields.py:223-272) restated.

    images_u8_320: (B,320,320,3) uint8 BGR (post-resize), sizes: list of (H,W).
    Returns a list of dicts with the reference's keys (tensors float32).
    taps: optional dict that receives the boundary tensors the engine's debug forward exposes (same names, NHWC): every MiT block output, c1..c4, ll,
    dec.<head>.conv0, pn.in / pn.stem / every ConvNeXt block output."""
    w = Weights(sd, "", dtype)
    x = torch.from_numpy(np.ascontiguousarray(images_u8_320)).permute(0, 3, 1, 2).to(torch.float32)
    x = normalise_input(x, dtype)
    feats = mit_b3(w.sub("backbone."), x, taps)
    ll = low_level_encoder(w.sub("ll_enc."), x)
    if taps is not None:
        for k, f in enumerate(feats):
            taps[f"c{k + 1}"] = f.permute(0, 2, 3, 1)
        taps["ll"] = ll.permute(0, 2, 3, 1)
    g = gravity_inference(w.sub("persformer_heads.gravity_head."), feats, ll, arch["gravity_cls"], taps)
    l = latitude_inference(w.sub("persformer_heads.latitude_head."), feats, ll, arch["latitude_cls"], taps)
    results = []
    for i, (H, W) in enumerate(sizes):
        gi, li = g[i].to(torch.float32), l[i].to(torch.float32)
        results.append(
            {
                "pred_gravity": gi,
                "pred_gravity_original": postprocess_gravity(gi, H, W, arch["gravity_cls"], arch["gravity_out"]),
                "pred_latitude": li,
                "pred_latitude_original": postprocess_latitude(li, H, W, arch["latitude_cls"], arch["latitude_out"]),
                "pred_latitude_original_mode": "deg",
            }
        )
    if arch["param_net"] is not None:
        p = param_net(w.sub("param_net."), g, l, arch, taps)
        raw = p.pop("raw")
        if "pred_general_vfov" not in p:
            p["pred_general_vfov"] = p["pred_vfov"]
        if "pred_rel_cx" not in p:
            p["pred_rel_cx"] = torch.zeros_like(p["pred_general_vfov"])
        if "pred_rel_cy" not in p:
            p["pred_rel_cy"] = torch.zeros_like(p["pred_general_vfov"])
        for i in range(len(results)):
            results[i].update({k: v[i].to(torch.float32) for k, v in p.items()})
            results[i]["_param_raw"] = raw[i].to(torch.float32)
    if stages:
        return results, {"feats": feats, "ll": ll, "gravity": g, "latitude": l}
    return results


def inference_batch(sd, arch, img_bgr_list, dtype=torch.float32):
    """inference_batch (perspectivefields.py:207-221): PIL resize each image, one forward."""
    sizes = [im.shape[:2] for im in img_bgr_list]
    x = np.stack([resize_to_net(im) for im in img_bgr_list])
    return forward(sd, arch, x, sizes, dtype)
