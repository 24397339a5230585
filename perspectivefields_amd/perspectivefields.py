"""Drop-in replacement for `perspective2d.PerspectiveFields` on MI355X.

Same surface as the reference class (perspective2d/perspectivefields.py:121-272):
`PerspectiveFields(version).eval().cuda()`, `.inference(img_bgr)`,
`.inference_batch(img_bgr_list)`, `.forward(batched_inputs)`, `.versions()`, attributes
`.version`, `.param_on`, `.cfg`, `.device`, module-level `model_zoo`; both entry points
run under no_grad and never mutate their inputs; the returned dicts have the reference's
keys, key order, shapes, dtypes (fp32) and units.

What differs by design: the nn.Module tree is replaced by one HIP engine (libpf_hip.so
through include/pf_hip.h); the checkpoint is validated strictly (the reference loads with
strict=False, :185,192); there is no network, so weights come from a local file, a
state_dict, or the seeded synthetic generator.  No CPU path exists: forward() on a CPU
device raises.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, List, Optional, Union

import numpy as np
import torch
from PIL import Image
from torch import nn

from . import config as _config
from .config import NET_H, NET_W, arch_of, get_cfg, model_zoo
from .engine import Engine, PfError
from .schema import checkpoint_schema, validate_state_dict
from .synth import synthetic_state_dict


class ResizeTransform:
    """HxW(xC) image -> new_h x new_w, aspect ratio not kept (reference: ResizeTransform.apply_image, perspectivefields.py:34-66): uint8 through PIL's antialiased
    BILINEAR (single-channel HxWx1 as mode "L"); any other dtype through torch's F.interpolate (no antialiasing, align_corners=False), as the reference does because
    "PIL only supports uint8".  Host-side preprocessing, as in the reference; inference()/inference_batch() can run the uint8 form on the GPU instead (device_resize)."""

    _PIL_TO_INTERPOLATE = {Image.NEAREST: "nearest", Image.BILINEAR: "bilinear", Image.BICUBIC: "bicubic"}

    def __init__(self, new_h: int, new_w: int, interp=None):
        self.new_h, self.new_w = new_h, new_w
        self.interp = Image.BILINEAR if interp is None else interp

    def apply_image(self, img: np.ndarray, interp=None) -> np.ndarray:
        assert len(img.shape) <= 4
        method = interp if interp is not None else self.interp
        if img.dtype == np.uint8:
            if len(img.shape) > 2 and img.shape[2] == 1:
                out = np.asarray(Image.fromarray(img[:, :, 0], mode="L").resize((self.new_w, self.new_h), method))
                return np.expand_dims(out, -1)
            return np.asarray(Image.fromarray(img).resize((self.new_w, self.new_h), method))
        if any(x < 0 for x in img.strides):
            img = np.ascontiguousarray(img)
        t = torch.from_numpy(img)
        shape = list(t.shape)
        shape_4d = shape[:2] + [1] * (4 - len(shape)) + shape[2:]
        t = t.view(shape_4d).permute(2, 3, 0, 1)  # hw(c) -> nchw
        mode = self._PIL_TO_INTERPOLATE[method]
        t = torch.nn.functional.interpolate(t, (self.new_h, self.new_w), mode=mode, align_corners=None if mode == "nearest" else False)
        shape[:2] = (self.new_h, self.new_w)
        return t.permute(2, 3, 0, 1).reshape(shape).numpy()  # nchw -> hw(c)


def _resolve_weights(version: str, weights) -> Dict[str, np.ndarray]:
    """weights: None | 'synthetic' | 'synthetic:<seed>' | path | state_dict | {'model': state_dict}."""
    if isinstance(weights, str) and weights.startswith("synthetic"):
        seed = int(weights.split(":", 1)[1]) if ":" in weights else 0
        return synthetic_state_dict(version, seed)
    if isinstance(weights, dict):
        sd = weights["model"] if "model" in weights and isinstance(weights["model"], dict) else weights
        return OrderedDict(sd)
    path = weights
    if path is None:
        fname = os.path.basename(model_zoo[version]["weights"])
        candidates = []
        if os.environ.get("PF_WEIGHTS_DIR"):
            candidates.append(os.path.join(os.environ["PF_WEIGHTS_DIR"], fname))
        candidates.append(os.path.join(torch.hub.get_dir(), "checkpoints", fname))
        path = next((c for c in candidates if os.path.exists(c)), None)
        if path is None:
            raise FileNotFoundError(
                f"no local checkpoint for '{version}' (looked for {candidates}); this build has no network access. "
                f"Download {model_zoo[version]['weights']} elsewhere and point PF_WEIGHTS_DIR at it, pass weights=<path|state_dict>, "
                "or weights='synthetic' for the seeded random checkpoint used by tests and benchmarks."
            )
    ckpt = torch.load(path, map_location="cpu", weights_only=True)  # a checkpoint is tensors in dicts: never unpickle arbitrary objects from a downloaded file
    return OrderedDict(ckpt["model"] if "model" in ckpt else ckpt)


def _window_limit(r) -> float:
    """upper end of the split-f16 window for one range record: 65504, except the attention operands, which carry extra powers of two inside the kernel
    (attn.hip: q x 8, k / v x 16) -- their window ends at 8188 / 4094 -- and the inputs of the Winograd convolutions (16376)"""
    name = r["name"]
    if "[winograd]" in name:   # the input transform of the Winograd convs adds four values (wino.hip) and does not clamp
        return 65504.0 / 4.0
    if name.startswith("attention") and name.endswith(" q"):
        return 8188.0
    if name.startswith("attention") and name.endswith(" kv"):
        return 4094.0
    return 65504.0


class PerspectiveFields(nn.Module):
    def __init__(self, version: str = "Paramnet-360Cities-edina-centered", weights=None, precision: str = "auto"):
        super().__init__()
        # 'fp32' is the parity mode: fp32-class contractions on the 2-way fp16 split -- full accuracy for activations in [2^-3, 65504], saturation beyond, an
        # absolute 2^-25 per element below (sb_split.h).  'fp32_bf16x6' is the exact bf16 split (no window, ~1.5x the MFMA work).  'auto' (the DEFAULT: a checkpoint
        # outside the window must not give wrong fields silently) decides between the two ONCE, on the first batch inference() / inference_batch() / forward() see:
        # a range-recording forward (pf_debug_forward_u8) and 'fp32_bf16x6' if any dense-layer input saturates or is all-tiny, 'fp32' otherwise (`self.precision`
        # then holds the decision, `self.precision_reason` why).  The decision is NOT final: in 'auto' every later forward is watched by the engine's saturation
        # counter (pf_set_saturation_counter: the producing kernels count outputs beyond their consumer's window), and a batch that moves it is re-run in
        # 'fp32_bf16x6', where the model then stays (a warning says so).  'fp32' pins the fast mode without the watch's host synchronisation.  No reduced-precision
        # mode is offered (include/pf_hip.h pf_set_precision says why).
        if precision not in ("auto", "fp32", "fp32_bf16x6"):
            raise ValueError(f"precision must be 'auto', 'fp32' or 'fp32_bf16x6', got '{precision}'")
        self._auto = precision == "auto"
        self.precision = precision
        cfg = get_cfg(version)  # KeyError on an unknown version, as the reference (:127)
        self.version = version
        self.param_on = model_zoo[version]["param"]
        self.cfg = cfg
        self.arch = arch_of(cfg)
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)
        self.input_format = cfg.INPUT.FORMAT
        self.aug = ResizeTransform(cfg.DATALOADER.RESIZE[0], cfg.DATALOADER.RESIZE[1])
        # True: inference()/inference_batch() upload the original uint8 image and run the (bit-identical) PIL resize
        # on the GPU instead of on a host core (1.7 ms/image/core); same bytes reach the network either way.
        self.device_resize = os.environ.get("PF_DEVICE_RESIZE", "0") == "1"
        self._engine: Optional[Engine] = None
        self._state: Dict[str, np.ndarray] = OrderedDict()
        self._init_weights(weights)

    # -------------------------------------------------------------- reference surface
    @property
    def device(self):
        return self.pixel_mean.device

    @staticmethod
    def versions():
        for key in model_zoo:
            print(f"{key}")
            print(f"   - {model_zoo[key]['description']}")

    def _init_weights(self, weights=None):
        sd = _resolve_weights(self.version, weights)
        # a checkpoint FILE is loaded like the reference does (strict=False: extra entries tolerated, with a warning);
        # state_dicts handed over in memory and the synthetic generator are held to the exact schema
        self.load_state_dict(sd, strict=not (weights is None or (isinstance(weights, str) and not weights.startswith("synthetic"))))

    def state_dict(self, *args, **kwargs):  # checkpoint-format view (host copies)
        return OrderedDict((k, torch.as_tensor(np.asarray(v))) for k, v in self._state.items())

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = state_dict["model"] if "model" in state_dict and isinstance(state_dict["model"], dict) else state_dict
        host = OrderedDict()
        for k, v in sd.items():
            host[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        # missing keys / wrong shapes are always fatal; keys this architecture does not have are fatal only when strict
        # (the reference loads with strict=False, :185,192: zoo files may carry extra entries) and are dropped otherwise
        extra = validate_state_dict(self.version, host, strict=strict)
        if extra:
            import warnings

            warnings.warn(f"PerspectiveFields.load_state_dict: ignoring {len(extra)} checkpoint entries this architecture does not use: {extra[:4]}")
            for k in extra:
                host.pop(k)
        self._state = host
        self._engine = None
        return self

    def _get_engine(self) -> Engine:
        dev = self.device
        if dev.type != "cuda":
            raise PfError(
                f"PerspectiveFields is on '{dev}': the MI355X engine has no CPU path. Call .cuda() / .to('cuda:N') first."
            )
        if self._engine is None or self._engine.device != dev:
            eng = Engine(self.arch["arch_id"], dev)
            eng.load_state_dict(self._state)
            eng.set_precision("fp32" if self.precision == "auto" else self.precision)  # always: the model's precision wins over any default of the library
            self._engine = eng
        return self._engine

    @torch.no_grad()
    def inference(self, img_bgr: np.ndarray) -> dict:
        return self.inference_batch([img_bgr])[0]

    @torch.no_grad()
    def inference_batch(self, img_bgr_list: List[np.ndarray]) -> List[dict]:
        """uint8 HxWx3 images (as cv2.imread returns) take the PIL path -- bit-identical bytes on the host or on the GPU; any other dtype (float images, 0..255) follows
        the reference's other branch (perspectivefields.py:48-66): F.interpolate on the host, then the float entry point of the engine (pf_forward_f32)."""
        batch, sizes = self._prepare_batch(img_bgr_list)
        return self._run(batch, sizes)

    @torch.no_grad()
    def inference_batch_with_params(self, img_bgr_list: List[np.ndarray]):
        """inference_batch plus the engine's raw (B, 8) camera-parameter tensor (include/pf_hip.h pf_forward_u8: d_params; None without a ParamNet) -- the per-image
        rows the multi-GPU path all-gathers (dist.ShardedPerspectiveFields).  An empty list gives ([], an empty (0, 8) tensor or None)."""
        if len(img_bgr_list) == 0:
            return [], (torch.zeros((0, 8), dtype=torch.float32, device=self.device) if self.param_on else None)
        batch, sizes = self._prepare_batch(img_bgr_list)
        lazy: list = []
        results = self._run(batch, sizes, lazy_params=lazy)   # joined forwards: the parameters are complete in stream order
        params = None
        if lazy:
            for res, pr in lazy:
                for r, extra in zip(res, self._param_dicts(pr)):
                    r.update(extra)
            params = lazy[0][1] if len(lazy) == 1 else torch.cat([pr for _, pr in lazy])
        return results, params

    def _prepare_batch(self, img_bgr_list: List[np.ndarray]):
        """host images -> (network input on the device, [(H, W)]): uint8 (B,320,320,3) through PIL (or the bit-identical device resize), float (B,3,320,320) otherwise"""
        if any(im.dtype != np.uint8 for im in img_bgr_list):
            sizes, chw = [], []
            for img_bgr in img_bgr_list:
                original = img_bgr[:, :, ::-1] if self.input_format == "RGB" else img_bgr
                sizes.append(tuple(int(v) for v in original.shape[:2]))
                r = self.aug.apply_image(np.ascontiguousarray(original))
                chw.append(torch.as_tensor(np.ascontiguousarray(r.astype("float32").transpose(2, 0, 1))))   # as the reference: image.astype("float32").transpose(2, 0, 1)
            return torch.stack(chw).to(self.device), sizes
        sizes, resized = [], []
        for img_bgr in img_bgr_list:
            original = img_bgr  # never mutated: apply_image returns a new array (reference copies, :196,211)
            if self.input_format == "RGB":
                original = original[:, :, ::-1]
            sizes.append(tuple(int(v) for v in original.shape[:2]))
            resized.append(np.ascontiguousarray(original) if self.device_resize else self.aug.apply_image(np.ascontiguousarray(original)))
        if self.device_resize:
            # one upload of all original images, then the whole batch resized in two launches per 32 images
            eng = self._get_engine()
            flat = torch.from_numpy(np.concatenate([im.reshape(-1) for im in resized])).to(self.device)
            views, o = [], 0
            for im in resized:
                views.append(flat[o:o + im.size].view(im.shape))
                o += im.size
            batch = torch.empty((len(resized), NET_H, NET_W, 3), dtype=torch.uint8, device=self.device)
            eng.resize_batch_into(views, batch)
        else:
            batch = torch.from_numpy(np.stack(resized)).to(self.device, non_blocking=False)  # uint8 (B,320,320,3)
        return batch, sizes

    # ------------------------------------------------------------------ debug forward (sanitizer / shadow-compare mode)
    @torch.no_grad()
    def debug_forward(self, img_bgr_list: List[np.ndarray], shadow: bool = True, ranges: bool = True):
        """The forward of `inference_batch` with the engine's debug sink (pf_debug_forward_u8): returns (results, taps, ranges).
        taps: {name: NHWC fp32 device tensor} of every block / stage boundary -- compare with `oracle.pf_oracle.forward(..., taps={})` to find the FIRST layer that
        is off; ranges: per dense layer input max |x|, rms, saturated / non-finite counts -- the first thing to run on a real checkpoint (see `check_range`)."""
        sizes = [tuple(int(v) for v in im.shape[:2]) for im in img_bgr_list]
        resized = [self.aug.apply_image(np.ascontiguousarray(im[:, :, ::-1] if self.input_format == "RGB" else im)) for im in img_bgr_list]
        batch = torch.from_numpy(np.stack(resized)).to(self.device)
        eng = self._get_engine()
        pg, pl, params, taps, rng = eng.forward_debug(batch, shadow=shadow, ranges=ranges)
        return self._assemble(eng, pg, pl, params, sizes), taps, rng

    def check_range(self, img_bgr_list: List[np.ndarray], verbose: bool = True) -> dict:
        """Range report of a checkpoint on real images.  The default precision ("fp32": 2-way fp16 split) represents |x| in [2^-3, 65504] with 22+ bits, SATURATES
        beyond 65504 and keeps only an absolute 2^-25 per element below 2^-3: a layer input with saturated elements, or whose rms is below 2^-5 (an all-tiny tensor), is
        outside the window -- use precision="fp32_bf16x6" (exact bf16 split, no window) for such a checkpoint.  Returns {"ok", "saturated", "tiny", "non_finite", "layers"}."""
        _, _, rng = self.debug_forward(img_bgr_list, shadow=False, ranges=True)
        sat = [r for r in rng if r["saturated"] > 0 or r["max_abs"] > _window_limit(r)]
        bad = [r for r in rng if r["non_finite"] > 0]
        tiny = [r for r in rng if 0.0 < r["rms"] < 2.0 ** -5]
        if verbose:
            print(f"{len(rng)} dense-layer inputs: max |x| {max((r['max_abs'] for r in rng), default=0.0):.4g}, smallest rms {min((r['rms'] for r in rng), default=0.0):.4g}; "
                  f"{len(sat)} with saturated elements, {len(tiny)} with rms < 2^-5, {len(bad)} with non-finite elements")
            for r in (sat + tiny + bad)[:20]:
                print(f"  {r['name']}: max |x| {r['max_abs']:.4g} rms {r['rms']:.4g} saturated {r['saturated']} non-finite {r['non_finite']}")
        return {"ok": not (sat or bad or tiny), "saturated": sat, "tiny": tiny, "non_finite": bad, "layers": rng}

    # ------------------------------------------------------------------ streaming
    _HOST_KEYS = ("pred_gravity", "pred_gravity_original", "pred_latitude", "pred_latitude_original")

    @torch.no_grad()
    def inference_stream(self, batches, to_host: bool = True, depth: int = 2, with_params: bool = False):
        """Pipelined inference over an iterable of image lists (see _inference_stream); leaves the engine's deferred-ParamNet mode off however the iteration ends.
        with_params: yield (results, raw (B, 8) camera-parameter tensor or None) pairs -- what dist.ShardedPerspectiveFields all-gathers."""
        try:
            yield from self._inference_stream(batches, to_host, depth, with_params)
        finally:
            eng = self._engine
            if eng is not None and getattr(eng, "defer_params", False):
                eng.set_defer_params(False)

    def _inference_stream(self, batches, to_host: bool = True, depth: int = 2, with_params: bool = False):
        """Pipelined inference over an iterable of image lists: yields one `inference_batch`-style result list per input
        batch, in order.  Three HIP streams overlap the stages of consecutive batches -- upload (pinned staging buffer,
        async H2D), compute (forward + post-process), download -- because the reference's callers move the fields to the
        host right after inference (demo/demo.py:55-58, 4.9 MB per 640x640 image).  With `to_host` the four field tensors
        of every result are pinned CPU tensors (filled by async D2H; complete when the batch is yielded); the ParamNet
        scalars stay 0-d device tensors as in `inference_batch`.  `depth` = batches in flight.
        With depth >= 2 the ParamNet branch of a batch runs on the engine's own stream beside the next batch's backbone (Engine.set_defer_params); a batch is yielded
        once ITS fields and ITS branch are complete (an event behind the branch, Engine.params_ready_event), with the scalar entries built after that."""
        dev = self.device
        if dev.type != "cuda":
            raise PfError(f"PerspectiveFields is on '{dev}': the MI355X engine has no CPU path. Call .cuda() first.")
        eng = self._get_engine()
        s_up, s_comp, s_down = (torch.cuda.Stream(device=dev) for _ in range(3))
        inflight = []
        # pinned upload staging: a ring of depth + 1 reusable buffers (a slot is rewritten only after its copy has completed)
        nslots = max(1, depth) + 1
        slots = [{"buf": None, "done": None} for _ in range(nslots)]
        nbatch = 0

        defer = self.param_on and depth >= 2
        if defer:
            eng.set_defer_params(True)
        keep_params = self.param_on and (defer or with_params)    # the scalar entries are built when a batch is finished, from its raw parameter tensors
        s_join = torch.cuda.Stream(device=dev) if keep_params else None  # waits for the deferred branches only (Engine.params_ready_event) / for the batch's own compute

        state = {"rerun_before": 0}   # batches with seq < this were issued in the fast mode before a window exit was seen: re-run unconditionally

        def finish(item):
            # With the deferred branch the camera parameters of `item` are written on the engine's own stream.  Wait for THAT branch only (an event recorded behind it on
            # s_join right after the forward was issued) -- not for the compute of the batch submitted after it, which would leave the GPU idle while the host prepares
            # the next batch -- and only then build the scalar entries: for ParamNetConvNextRegress they are arithmetic on `params` (factors), which must not be launched
            # before the branch has written them.
            if defer:
                item["params_done"].synchronize()
            item["done"].synchronize()
            rerun = False
            if item["sat"] is not None and item["fast"]:
                # precision='auto': did this batch leave the split-f16 window?  Two looks at the counter, both behind finished work (reading them costs nothing):
                # the snapshots behind each chunk's forward on the compute stream, and -- deferred branch -- one behind the BRANCH on s_join: the ParamNet kernels of batch
                # i run after snapshot i was taken (beside batch i + 1's backbone), so without it their increments would be charged to batch i + 1, or to nobody for the
                # last batch.  The counter is global: an increment seen here may belong to a later batch that is already running -- so once it moves, EVERY batch that
                # was issued in the fast mode (at most `depth` of them) is re-run in the exact mode when its turn comes, not only this one.
                snaps = [snap for snap, _, _, _ in item["sat"]] + ([item["sat_branch"]] if item["sat_branch"] is not None else [])
                moved = [self._left_window(eng, snap) for snap in snaps]
                if any(moved):
                    state["rerun_before"] = max(state["rerun_before"], nbatch)
                rerun = item["seq"] < state["rerun_before"]

            def out(res, lazy):
                if not with_params:
                    return res
                prs = [pr for _, pr in (lazy or [])]
                return res, (None if not prs else prs[0] if len(prs) == 1 else torch.cat(prs))

            if rerun:
                lz = [] if keep_params else None
                with torch.cuda.stream(s_comp):
                    new = self._rerun_exact(eng, item["batch"], item["sizes"], lz)   # joined forwards: the parameters are complete in stream order
                    for res, params in (lz or []):
                        for r, extra in zip(res, self._param_dicts(params)):
                            r.update(extra)
                    if with_params and lz:
                        lz = [(None, torch.cat([pr for _, pr in lz]))]
                s_comp.synchronize()
                for r, n in zip(item["results"], new):
                    r.clear()
                    r.update({k: (v.cpu() if (to_host and k in self._HOST_KEYS) else v) for k, v in n.items()})
                return out(item["results"], lz)
            if keep_params:
                with torch.cuda.stream(s_join):  # NOT the compute stream: it already holds the next batch's forward
                    if not defer:
                        s_join.wait_event(item["comp_done"])
                    for res, params in item["lazy"]:
                        for r, extra in zip(res, self._param_dicts(params)):
                            r.update(extra)
                    if with_params and len(item["lazy"]) > 1:
                        item["lazy"] = [(None, torch.cat([pr for _, pr in item["lazy"]]))]
                s_join.synchronize()
            return out(item["results"], item["lazy"])

        for imgs in batches:
            sizes, resized = [], []
            for img_bgr in imgs:
                original = img_bgr[:, :, ::-1] if self.input_format == "RGB" else img_bgr
                if original.dtype != np.uint8:
                    raise TypeError("PerspectiveFields expects uint8 BGR images (as cv2.imread returns)")
                sizes.append(tuple(int(v) for v in original.shape[:2]))
                resized.append(np.ascontiguousarray(original) if self.device_resize else self.aug.apply_image(np.ascontiguousarray(original)))
            slot = slots[nbatch % nslots]
            nbatch += 1
            if slot["done"] is not None:
                slot["done"].synchronize()
            nbytes = sum(int(im.size) for im in resized)
            if slot["buf"] is None or slot["buf"].numel() < nbytes:
                slot["buf"] = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
            host = slot["buf"]
            host_np = host.numpy()  # shares the pinned memory
            offs, o = [], 0
            for im in resized:
                np.copyto(host_np[o:o + im.size], im.reshape(-1))
                offs.append(o)
                o += int(im.size)
            with torch.cuda.stream(s_up):
                dev_in = host[:nbytes].to(dev, non_blocking=True)  # one H2D per batch
                if self.device_resize:
                    batch = torch.empty((len(resized), NET_H, NET_W, 3), dtype=torch.uint8, device=dev)
                    eng.resize_batch_into([dev_in[offs[i]:offs[i] + im.size].view(im.shape) for i, im in enumerate(resized)], batch)
                else:
                    batch = dev_in.view(len(resized), NET_H, NET_W, 3)
                up_done = torch.cuda.Event()
                up_done.record(s_up)
            slot["done"] = up_done
            batch.record_stream(s_comp)
            s_comp.wait_event(up_done)
            lazy = [] if keep_params else None
            sat = [] if self._auto else None
            with torch.cuda.stream(s_comp):
                results = self._run(batch, sizes, lazy_params=lazy, sat_out=sat)
                comp_done = torch.cuda.Event()
                comp_done.record(s_comp)
            params_done = eng.params_ready_event(s_join) if defer else None
            sat_branch = None
            if defer and sat is not None and sat:   # the counter as of the END of this batch's ParamNet branch (s_join is behind it)
                with torch.cuda.stream(s_join):
                    sat_branch = eng.saturation_snapshot()
            done = comp_done
            if to_host:
                s_down.wait_event(comp_done)
                total = sum(r[k].numel() for r in results for k in self._HOST_KEYS)
                pinned = torch.empty(total, dtype=torch.float32, pin_memory=True)  # one pinned block per batch, sliced into views
                with torch.cuda.stream(s_down):
                    o = 0
                    for r in results:
                        for k in self._HOST_KEYS:
                            src = r[k]
                            src.record_stream(s_down)
                            dst = pinned[o:o + src.numel()].view(src.shape)
                            dst.copy_(src, non_blocking=True)
                            r[k] = dst
                            o += src.numel()
                    done = torch.cuda.Event()
                    done.record(s_down)
            inflight.append({"results": results, "done": done, "comp_done": comp_done, "params_done": params_done, "lazy": lazy, "sat": sat, "sat_branch": sat_branch,
                             "batch": batch, "sizes": sizes, "seq": nbatch - 1, "fast": self.precision == "fp32"})
            if len(inflight) >= max(1, depth):
                yield finish(inflight.pop(0))
        while inflight:
            yield finish(inflight.pop(0))
        if defer:
            with torch.cuda.stream(s_comp):
                eng.set_defer_params(False)

    def fields_from_prediction(self, pred: dict, height: int, width: int):
        """Perspective fields implied by the ParamNet scalars of one inference() result (see fields_from_params)."""
        if not self.param_on:
            raise PfError(f"'{self.version}' has no ParamNet: there are no camera parameters to synthesise fields from")
        return fields_from_params(pred["pred_roll"], pred["pred_pitch"], pred["pred_rel_focal"], pred["pred_rel_cx"], pred["pred_rel_cy"],
                                  height, width, mode="deg")

    def forward(self, batched_inputs) -> List[dict]:
        """batched_inputs: list of {"image": (3,320,320) float BGR 0..255, "height", "width"} (reference :223-272)."""
        with torch.no_grad():
            imgs = torch.stack([x["image"].to(self.device, dtype=torch.float32) for x in batched_inputs])
            sizes = [(int(x.get("height")), int(x.get("width"))) for x in batched_inputs]
            return self._run(imgs, sizes)

    # -------------------------------------------------------------------- engine path
    # images per engine forward: longer lists are processed in chunks (the reference accepts any list length; one
    # pf_forward call is limited to PF_MAX_BATCH = 81 images by its 32-bit activation offsets, include/pf_hip.h)
    MAX_CHUNK = 64

    def _run(self, batch, sizes, lazy_params=None, sat_out=None) -> List[dict]:
        """lazy_params: a list -> the ParamNet entries are NOT added to the result dicts here; (results of the chunk, raw (B,8) params) pairs are appended instead and
        the caller adds them once the parameters are complete (inference_stream with the deferred branch).  sat_out: a list -> the saturation watch of 'auto' is not
        read here (that would synchronise); (snapshot, batch, sizes, results) is appended for the caller to look at when the batch has finished."""
        eng = self._get_engine()
        chunk = max(1, min(int(os.environ.get("PF_MAX_CHUNK", self.MAX_CHUNK)), eng.max_batch))
        if len(sizes) > chunk:
            out: List[dict] = []
            for i0 in range(0, len(sizes), chunk):
                out.extend(self._run(batch[i0:i0 + chunk], sizes[i0:i0 + chunk], lazy_params, sat_out))
            return out
        if self.precision == "auto":  # first batch: look at the weights' static bounds and at the activations this checkpoint produces, then settle on a precision
            probe = batch if batch.dtype == torch.uint8 else batch.permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8)  # forward(): (B,3,320,320) float -> the u8 NHWC form of the debug entry
            _, _, _, _, rng = eng.forward_debug(probe.contiguous(), shadow=False, ranges=True)
            outside = [r for r in rng if r["saturated"] > 0 or r["max_abs"] > _window_limit(r) or r["non_finite"] > 0 or 0.0 < r["rms"] < 2.0 ** -5]
            static_max = eng.static_window_max()
            self.precision = "fp32_bf16x6" if (outside or static_max > 65504.0) else "fp32"
            if outside:
                self.precision_reason = (f"{len(outside)} of {len(rng)} dense-layer inputs outside the split-f16 window, first: {outside[0]['name']} "
                                         f"(max |x| {outside[0]['max_abs']:.4g}, rms {outside[0]['rms']:.4g})")
            elif static_max > 65504.0:
                self.precision_reason = f"a weights-only bound of an unwatched tensor (LayerNorm output / fused-MLP hidden map) reaches {static_max:.4g} > 65504"
            else:
                self.precision_reason = f"all {len(rng)} dense-layer inputs inside the split-f16 window (static bounds <= {static_max:.4g}); later batches are watched by the saturation counter"
            eng.set_precision(self.precision)
            eng.sat_seen = int(eng.saturation_snapshot())
        watch = self._auto and self.precision == "fp32"
        pg, pl, params = eng.forward(batch)
        snap = eng.saturation_snapshot() if watch else None
        results = self._assemble(eng, pg, pl, params, sizes, lazy_params)
        if watch:
            if sat_out is not None:
                sat_out.append((snap, batch, sizes, results))   # inference_stream looks when the batch is finished
            elif self._left_window(eng, snap):
                if lazy_params is not None and params is not None:
                    lazy_params.pop()
                results = self._rerun_exact(eng, batch, sizes, lazy_params)
        return results

    def _left_window(self, eng, snap) -> bool:
        """True when the forward behind `snap` moved the saturation counter (reading it synchronises with that forward)."""
        cnt = int(snap)
        moved = cnt != eng.sat_seen
        eng.sat_seen = cnt
        return moved

    def _rerun_exact(self, eng, batch, sizes, lazy_params=None):
        """Re-run of a batch that left the split-f16 window, in the exact mode (where the model then stays).  Always JOINED forwards: with the deferred ParamNet branch
        on, the branch is switched off around the re-run (which also puts the current stream behind a pending branch of a later batch), so that the scalar entries built
        right here read finished parameters; chunked like _run (a stream batch may exceed the engine's per-forward limit)."""
        import warnings

        if self.precision != "fp32_bf16x6":
            self.precision = "fp32_bf16x6"
            self.precision_reason = "a later batch left the split-f16 window (saturation counter moved): re-run and continuing in the exact bf16 split"
            warnings.warn("PerspectiveFields(precision='auto'): " + self.precision_reason)
            eng.set_precision(self.precision)
        deferred = bool(getattr(eng, "defer_params", False))
        if deferred:
            eng.set_defer_params(False)
        try:
            chunk = max(1, min(int(os.environ.get("PF_MAX_CHUNK", self.MAX_CHUNK)), eng.max_batch))
            out: List[dict] = []
            for i0 in range(0, len(sizes), chunk):
                pg, pl, params = eng.forward(batch[i0:i0 + chunk])
                out.extend(self._assemble(eng, pg, pl, params, sizes[i0:i0 + chunk], lazy_params))
        finally:
            if deferred:
                eng.set_defer_params(True)
        return out

    def _assemble(self, eng, pg, pl, params, sizes, lazy_params=None) -> List[dict]:
        results = []
        fields = eng.postprocess_batch(pg, pl, sizes)  # the reference's per-image post-process loop as one launch
        for i, (h, w) in enumerate(sizes):
            up, lat = fields[i]
            results.append(
                {
                    "pred_gravity": pg[i],
                    "pred_gravity_original": up,
                    "pred_latitude": pl[i],
                    "pred_latitude_original": lat,
                    "pred_latitude_original_mode": "deg",
                }
            )
        if params is not None:
            if lazy_params is not None:
                lazy_params.append((results, params))
            else:
                for i, extra in enumerate(self._param_dicts(params)):
                    results[i].update(extra)
        return results

    def _param_dicts(self, params) -> List[dict]:
        """(B,8) engine output -> the reference's per-image scalar dict entries
        (param_network.py:54-69 / 199-221 and perspectivefields.py:260-271)."""
        B = params.shape[0]
        if self.arch["param_net"] == "ParamNet":
            zeros = torch.zeros(B, dtype=torch.float32, device=params.device)
            cols = OrderedDict(
                pred_roll=params[:, 0], pred_pitch=params[:, 1], pred_vfov=params[:, 2], pred_rel_focal=params[:, 3],
                pred_general_vfov=params[:, 2], pred_rel_cx=zeros, pred_rel_cy=zeros,
            )
            return [{k: v[i] for k, v in cols.items()} for i in range(B)]
        # ParamNetConvNextRegress: factors per predicted parameter; rel_focal from general_vfov in closed form -- on the device for the zoo's output order (column 5 of the
        # engine's (B, 8) output, paramnet_scalars_kernel: no host round trip on the hot path), on the host for any other order of PREDICT_PARAMS
        factors = {"roll": 90.0, "pitch": 90.0, "vfov": 90.0, "rel_focal": 1.0, "rel_cx": 1.0, "rel_cy": 1.0, "general_vfov": 90.0}
        cols = OrderedDict()
        for j, key in enumerate(self.arch["predict_params"]):
            cols["pred_" + key] = params[:, j] * factors[key]
        if "pred_rel_focal" not in cols:
            if list(self.arch["predict_params"]) == list(self._DEVICE_FOCAL_ORDER):
                cols["pred_rel_focal"] = params[:, 5]
            else:
                cols["pred_rel_focal"] = torch.from_numpy(
                    general_vfov_to_focal(
                        cols["pred_rel_cx"].double().cpu().numpy(), cols["pred_rel_cy"].double().cpu().numpy(),
                        cols["pred_general_vfov"].double().cpu().numpy(),
                    ).astype(np.float32)
                )
        return [{k: v[i] for k, v in cols.items()} for i in range(B)]

    _DEVICE_FOCAL_ORDER = ("roll", "pitch", "general_vfov", "rel_cx", "rel_cy")   # every ParamNetConvNextRegress entry of the zoo (config/paramnet_*_rpfpp.yaml:32-37)


def fields_from_params(roll, pitch, rel_focal, rel_cx=0.0, rel_cy=0.0, height=None, width=None, mode="deg", device=None):
    """Camera parameters -> (up field (2,H,W) unit vectors, latitude map (H,W) in degrees) on the GPU: the step the
    reference's demos run right after inference (utils/utils.py:325-381 -> PanoCam.get_up_general / get_lat_general,
    utils/panocam.py:451-556), e.g. to compare the ParamNet output with the predicted fields.  Same layout and units as
    `pred_gravity_original` / `pred_latitude_original`.  Arguments may be Python floats or 0-d tensors (the `pred_*`
    entries of an inference dict: they stay on the device, no host round trip); angles in degrees unless mode="rad"."""
    import ctypes  # noqa: F401

    from .engine import _check, _stream_ptr, load_library

    if height is None or width is None:
        raise ValueError("fields_from_params needs the output size (height, width)")
    dev = None
    for v in (roll, pitch, rel_focal, rel_cx, rel_cy):
        if torch.is_tensor(v) and v.is_cuda:
            dev = v.device
    if dev is None:
        dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise PfError("fields_from_params runs on the GPU only (no CPU path)")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    t = [torch.as_tensor(v, dtype=torch.float64).to(dev).reshape(()) for v in (roll, pitch, rel_focal, rel_cx, rel_cy)]
    if mode == "deg":
        t[0], t[1] = torch.deg2rad(t[0]), torch.deg2rad(t[1])
    elif mode != "rad":
        raise ValueError("mode must be 'deg' or 'rad'")
    cam = torch.stack(t).to(torch.float32).contiguous()
    up = torch.empty((2, int(height), int(width)), dtype=torch.float32, device=dev)
    lat = torch.empty((int(height), int(width)), dtype=torch.float32, device=dev)
    lib = load_library()
    with torch.cuda.device(dev):
        _check(lib.pf_fields_from_params(dev.index, cam.data_ptr(), int(height), int(width), up.data_ptr(), lat.data_ptr(), _stream_ptr()),
               None, "pf_fields_from_params")
    return up, lat


def general_vfov_to_focal(rel_cx, rel_cy, gvfov_deg):
    """Relative focal length from the general vertical FoV (reference: utils/utils.py:47-91 with h=1,
    degree=True).  With u = f^2 + cx^2 + cy^2 + 1/4 the reference's equation
    cos(gvfov) = (u - 1/2) / sqrt(u^2 - cy^2) is a quadratic in u, solved here in closed form
    (the reference runs scipy.fsolve from 1.5 on the same equation and returns |f|)."""
    cx = np.asarray(rel_cx, dtype=np.float64)
    cy = np.asarray(rel_cy, dtype=np.float64)
    c = np.cos(np.radians(np.asarray(gvfov_deg, dtype=np.float64)))
    s2 = 1.0 - c * c
    disc = np.sqrt(np.maximum(1.0 - 4.0 * s2 * (c * c * cy * cy + 0.25), 0.0))
    # root selection: cos > 0 needs u > 1/2 -> '+' root; cos < 0 (gvfov > 90 deg) -> '-' root
    u = np.where(c >= 0, (1.0 + disc), (1.0 - disc)) / (2.0 * s2)
    f2 = u - cy * cy - 0.25 - cx * cx
    return np.sqrt(np.abs(f2))
