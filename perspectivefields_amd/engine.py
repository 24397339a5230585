"""ctypes binding of libpf_hip.so (include/pf_hip.h) -- the only compute path.

PyTorch is used for device memory and stream ownership only: tensors are allocated with
torch, their raw device pointers and torch's current HIP stream are handed to the C ABI.
There is NO fallback: a missing library, a missing GPU or a non-gfx950 device raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import numpy as np

from .build import LIB as LIB_PATH  # lib/libpf_hip.so (lib_tune/, lib_lo/ for the build variants, see build.py)

_HERE = os.path.dirname(os.path.abspath(__file__))
TILE_TABLE = os.path.join(_HERE, "tuned", "gfx950_tiles.txt")  # per-shape tile choices measured on MI355X (scripts/gen_tile_table.py)
NET = 320
PARAMS_STRIDE = 8

PF_OK = 0
_STATUS = {0: "PF_OK", -1: "PF_ERR_ARG", -2: "PF_ERR_DEVICE", -3: "PF_ERR_WEIGHTS", -4: "PF_ERR_WORKSPACE"}

# every symbol include/pf_hip.h declares: name -> (restype, argtypes)
_c = ctypes
_P = _c.c_void_p
_SIGNATURES = {
    "pf_version": (_c.c_char_p, []),
    "pf_last_error": (_c.c_char_p, [_P]),
    "pf_create": (_c.c_int, [_c.POINTER(_P), _c.c_int, _c.c_int]),
    "pf_destroy": (_c.c_int, [_P]),
    "pf_load_tensor": (_c.c_int, [_P, _c.c_char_p, _P, _c.POINTER(_c.c_int64), _c.c_int]),
    "pf_finalize_weights": (_c.c_int, [_P]),
    "pf_output_info": (_c.c_int, [_P, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "pf_max_batch": (_c.c_int, []),
    "pf_workspace_bytes": (_c.c_size_t, [_P, _c.c_int]),
    "pf_forward_u8": (_c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _P, _c.c_size_t, _P]),
    "pf_forward_f32": (_c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _P, _c.c_size_t, _P]),
    "pf_set_saturation_counter": (_c.c_int, [_P, _P]),
    "pf_static_window_max": (_c.c_int, [_P, _c.POINTER(_c.c_float)]),
    "pf_set_defer_params": (_c.c_int, [_P, _c.c_int]),
    "pf_join_params": (_c.c_int, [_P, _P]),
    "pf_forward_u8_graph": (_c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _P, _c.c_size_t, _P]),
    "pf_resize_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "pf_resize_bilinear_u8": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _P, _P, _c.c_size_t, _P]),
    "pf_resize_batch_u8": (_c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _c.c_size_t, _P]),
    "pf_autotune": (_c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _P, _c.c_size_t, _P]),
    "pf_is_tuned": (_c.c_int, [_P, _c.c_int]),
    "pf_autotune_workspace_bytes": (_c.c_size_t, [_P, _c.c_int]),
    "pf_load_tile_table": (_c.c_int, [_P, _c.c_char_p]),
    "pf_save_tile_table": (_c.c_int, [_P, _c.c_char_p]),
    "pf_build_digest": (_c.c_char_p, []),
    "pf_set_precision": (_c.c_int, [_P, _c.c_int]),
    "pf_postprocess": (_c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _P, _P, _P, _c.c_size_t, _P]),
    "pf_postprocess_batch": (_c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _P, _P, _c.c_size_t, _P]),
    "pf_fields_from_params": (_c.c_int, [_c.c_int, _P, _c.c_int, _c.c_int, _P, _P, _P]),
    "pf_profile_begin": (_c.c_int, [_P, _c.c_uint]),
    "pf_profile_pause": (_c.c_int, [_P]),
    "pf_profile_end": (_c.c_int, [_P, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.POINTER(_c.c_long), _c.c_int]),
    "pf_profile_records": (_c.c_int, [_P, _c.c_int, _c.POINTER(_c.c_int), _c.POINTER(_c.c_double), _c.POINTER(_c.c_float), _c.POINTER(_c.c_int)]),
    "pf_profile_phases": (_c.c_int, [_P, _c.c_int, _c.POINTER(_c.c_double)]),
    "pf_debug_tap_bytes": (_c.c_size_t, [_P, _c.c_int]),
    "pf_debug_forward_u8": (_c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _P, _c.c_size_t, _c.c_int, _P, _c.c_size_t, _P]),
    "pf_debug_taps": (_c.c_int, [_P, _c.c_int, _P, _c.POINTER(_c.c_longlong), _c.POINTER(_c.c_int)]),
    "pf_debug_ranges": (_c.c_int, [_P, _c.c_int, _P, _c.POINTER(_c.c_longlong), _c.POINTER(_c.c_float)]),
    "pf_op_conv2d": (_c.c_int, [_c.c_int, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P,
                                _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _c.c_int,
                                _c.c_int, _c.c_int, _P, _P, _c.c_long, _P, _c.c_long, _P, _c.c_long, _c.c_int, _P]),
    "pf_op_conv2d_bench": (_c.c_int, [_c.c_int] * 12 + [_c.POINTER(_c.c_float)]),
    "pf_op_split_bf16": (_c.c_int, [_c.c_int, _P, _c.c_long, _P, _c.c_long, _P]),
    "pf_op_merge_bf16": (_c.c_int, [_c.c_int, _P, _c.c_long, _c.c_long, _P, _P]),
    "pf_op_dwconv3x3_bench": (_c.c_int, [_c.c_int] * 7 + [_c.POINTER(_c.c_float)]),
    "pf_op_layernorm": (_c.c_int, [_c.c_int, _P, _P, _P, _P, _c.c_long, _c.c_int, _c.c_float, _P, _c.c_long, _P]),
    "pf_op_dwconv3x3_gelu": (_c.c_int, [_c.c_int, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_long, _P]),
    "pf_op_mit_mlp": (_c.c_int, [_c.c_int, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _c.c_float, _P, _P, _P, _P, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_cnx_mlp": (_c.c_int, [_c.c_int, _P, _P, _c.c_long, _c.c_int, _P, _P, _P, _P, _c.c_float, _P, _P, _P, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_rb_linear": (_c.c_int, [_c.c_int, _P, _c.c_long, _c.c_int, _c.c_int, _P, _P, _P, _P, _c.c_float, _c.c_int, _c.c_int, _P, _P, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_rb_proj_fc1": (_c.c_int, [_c.c_int, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _c.c_float, _P, _P, _P, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_mit_attn64": (_c.c_int, [_c.c_int, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P, _c.c_float, _P, _P, _P, _P, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_stem7x7": (_c.c_int, [_c.c_int, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _c.c_int, _P, _P, _c.c_float, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_thin128": (_c.c_int, [_c.c_int, _P, _c.c_long, _P, _P, _P, _P, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_rb_srkv": (_c.c_int, [_c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _c.c_float, _P, _P, _P, _P, _c.c_float, _P, _P, _P, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_linear_ln": (_c.c_int, [_c.c_int, _P, _c.c_long, _c.c_int, _P, _P, _P, _P, _c.c_float, _c.c_int, _c.c_int, _P, _c.c_int, _P, _c.c_int, _P]),
    "pf_op_dwconv3x3_gelu_cfg": (_c.c_int, [_c.c_int, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_long, _c.c_int, _P]),
    "pf_op_dwconv7x7": (_c.c_int, [_c.c_int, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P]),
    "pf_op_dwconv7x7_cfg": (_c.c_int, [_c.c_int, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P]),
    "pf_op_dwconv7x7_bench": (_c.c_int, [_c.c_int] * 10 + [_c.POINTER(_c.c_float)]),
    "pf_op_sr_attention": (_c.c_int, [_c.c_int, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_long, _P]),
    "pf_op_sr_attention_variant": (_c.c_int, [_c.c_int, _c.c_int, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_float), _P]),
    "pf_op_upsample2x": (_c.c_int, [_c.c_int, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_long, _P]),
    "pf_op_num_conv_tiles": (_c.c_int, []),
    "pf_op_conv_tile_name": (_c.c_char_p, [_c.c_int]),
}

_lib = None


class PfError(RuntimeError):
    pass


def load_library(path: Optional[str] = None):
    """dlopen libpf_hip.so and bind every declared symbol.  Raises if the library has
    not been built (run `python -m perspectivefields_amd.build` or __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    want = None
    if path is None:
        # The library must match the sources next to it: a stale .so after a csrc / header change would run ctypes calls with
        # mismatched signatures.  Rebuild in-tree when hipcc is present (one process builds, the others wait on the lock);
        # otherwise fail loudly.  PF_SKIP_DIGEST_CHECK=1 disables the comparison (deployments without the sources).
        from . import build as _build

        stamp = os.path.join(os.path.dirname(LIB_PATH), "libpf_hip.digest")
        try:
            want = None if os.environ.get("PF_SKIP_DIGEST_CHECK") == "1" else _build._digest()
        except OSError:
            want = None  # sources not shipped
        have = open(stamp).read() if os.path.exists(stamp) else None
        if not os.path.exists(p) or (want is not None and have != want):
            try:
                import fcntl

                os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
                with open(os.path.join(os.path.dirname(LIB_PATH), ".build.lock"), "w") as lk:
                    fcntl.flock(lk, fcntl.LOCK_EX)
                    _build.build(verbose=False)  # no-op when another process has just built it
            except Exception as e:  # no hipcc / build failure: fail loudly
                what = "not found" if not os.path.exists(p) else "is stale (built from different sources)"
                raise PfError(
                    f"{p} {what} and could not be built ({e}); the HIP extension is required, there is no CPU fallback. "
                    "Build it with `python -m perspectivefields_amd.build` (needs hipcc)."
                ) from e
    if not os.path.exists(p):
        raise PfError(f"{p} not found: the HIP extension is not built and there is no CPU fallback.")
    lib = ctypes.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI drift between header and library
        fn.restype = res
        fn.argtypes = args
    if path is None:
        got = lib.pf_build_digest().decode()
        if want is not None and got != want:
            raise PfError(f"{p} was built from different sources (library digest {got[:12]}, sources {want[:12]}): rebuild with `python -m perspectivefields_amd.build --force`")
        _lib = lib
    return lib


def declared_symbols():
    return list(_SIGNATURES)


def _stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(rc: int, handle=None, what: str = ""):
    if rc != PF_OK:
        lib = load_library()
        msg = lib.pf_last_error(handle)
        raise PfError(f"{what}: {_STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")


def _dev_index(device) -> int:
    import torch

    d = torch.device(device)
    if d.type != "cuda":
        raise PfError(f"the PerspectiveFields HIP engine needs a GPU device, got '{d}' (no CPU fallback)")
    return d.index if d.index is not None else torch.cuda.current_device()


class Engine:
    """One network (architecture + weights) resident on one GPU."""

    def __init__(self, arch_id: int, device):
        import torch

        if not torch.cuda.is_available():
            raise PfError("no GPU visible to PyTorch-ROCm: the HIP engine cannot run (no CPU fallback)")
        self.lib = load_library()
        self.device = torch.device("cuda", _dev_index(device))
        self.arch_id = arch_id
        self._h = ctypes.c_void_p()
        _check(self.lib.pf_create(ctypes.byref(self._h), self.device.index, arch_id), None, "pf_create")
        self._ws = None
        self._scratch_stream = {}  # scratch name -> the torch stream of its last user (_order_scratch)
        self._finalized = False
        self.precision = "fp32"
        self.max_batch = int(self.lib.pf_max_batch())
        # Opt-in: batches up to this size replay a captured hipGraph (pf_forward_u8_graph).  Off by default -- measured on MI355X
        # (profiles/r02_latency.md): a batch-1 forward is bound by the GPU-side latency of its ~430 dependent small kernels
        # (6.4 ms eager, 6.8 ms replayed), not by host launch cost, so the replay buys nothing there.
        self.graph_max_batch = int(os.environ.get("PF_GRAPH_MAX_BATCH", "0"))
        self.defer_params = False   # set_defer_params()
        self._deferred = []
        self._graph_bufs = {}
        # always-on saturation watch of the split-f16 mode (pf_set_saturation_counter): a device counter the producing kernels add to when an output leaves its
        # consumer's window; read in stream order with saturation_snapshot()
        with torch.cuda.device(self.device):
            self._sat = torch.zeros(1, dtype=torch.int32, device=self.device)
        _check(self.lib.pf_set_saturation_counter(self._h, self._sat.data_ptr()), self._h, "pf_set_saturation_counter")
        self.sat_seen = 0   # counter value up to which the caller has looked
        g, l, p = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.lib.pf_output_info(self._h, ctypes.byref(g), ctypes.byref(l), ctypes.byref(p))
        self.gravity_channels, self.latitude_channels, self.param_outputs = g.value, l.value, p.value

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.pf_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    def load_state_dict(self, state_dict: Dict[str, object]):
        """Strict load of a reference-format state_dict (values: numpy arrays or torch tensors)."""
        for key, val in state_dict.items():
            arr = val.detach().cpu().numpy() if hasattr(val, "detach") else np.asarray(val)
            if key.endswith("num_batches_tracked"):
                arr = np.zeros((), dtype=np.float32)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (ctypes.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _check(
                self.lib.pf_load_tensor(self._h, key.encode(), arr.ctypes.data_as(ctypes.c_void_p), shape, arr.ndim),
                self._h, f"pf_load_tensor({key})",
            )
        _check(self.lib.pf_finalize_weights(self._h), self._h, "pf_finalize_weights")
        self._finalized = True
        table = os.environ.get("PF_TILE_TABLE", TILE_TABLE)
        if table and os.path.exists(table):
            self.tile_table_entries = self.lib.pf_load_tile_table(self._h, table.encode())

    def autotune(self, batch: int, save_to: Optional[str] = None) -> None:
        """Explicit one-time tile tuning for a batch size (pf_autotune: times every tile configuration of every conv /
        GEMM shape on this device, ~1-2 s, synchronises).  Not needed for the shipped table's batch sizes."""
        import torch

        with torch.cuda.device(self.device):
            x = torch.randint(0, 256, (batch, NET, NET, 3), dtype=torch.uint8, device=self.device)
            pg = torch.empty((batch, self.gravity_channels, NET, NET), dtype=torch.float32, device=self.device)
            pl = torch.empty((batch, self.latitude_channels, NET, NET), dtype=torch.float32, device=self.device)
            params = torch.empty((batch, PARAMS_STRIDE), dtype=torch.float32, device=self.device) if self.param_outputs else None
            ws = self._workspace(int(self.lib.pf_autotune_workspace_bytes(self._h, batch)))
            rc = self.lib.pf_autotune(self._h, batch, x.data_ptr(), pg.data_ptr(), pl.data_ptr(),
                                      params.data_ptr() if params is not None else None, ws.data_ptr(), ws.numel(), _stream_ptr())
        _check(rc, self._h, "pf_autotune")
        if save_to:
            _check(min(0, self.lib.pf_save_tile_table(self._h, save_to.encode())), self._h, "pf_save_tile_table")

    PRECISIONS = {"fp32": 0, "fp32_bf16x6": 3}

    def set_precision(self, mode: str):
        """Arithmetic of the dense contractions: 'fp32' (default, the parity mode: 2-way fp16 split, three MFMAs per
        product) or 'fp32_bf16x6' (exact 3-way bf16 split, six MFMAs: fp32-accurate for any input range).  No reduced-precision mode is offered
        (pf_set_precision in include/pf_hip.h says why)."""
        if mode not in self.PRECISIONS:
            raise PfError(f"unknown precision '{mode}' (expected one of {sorted(self.PRECISIONS)})")
        _check(self.lib.pf_set_precision(self._h, self.PRECISIONS[mode]), self._h, "pf_set_precision")
        self.precision = mode
        self._graph_bufs = {}  # captured graphs and their workspace were sized for the previous mode (the workspace differs between the schemes)

    def saturation_snapshot(self):
        """The saturation counter as of the work issued so far on the current stream (a 1-element device tensor: reading it synchronises).  It only grows; a forward
        that leaves it unchanged had every dense-layer input inside the split-f16 window."""
        return self._sat.clone()

    def static_window_max(self) -> float:
        """Largest static (weights-only) bound of a tensor no kernel can watch, scaled to the 65504 window (pf_static_window_max): > 65504 = may leave the window."""
        v = ctypes.c_float()
        _check(self.lib.pf_static_window_max(self._h, ctypes.byref(v)), self._h, "pf_static_window_max")
        return float(v.value)

    # ---------------------------------------------------------------- forward
    def workspace_bytes(self, batch: int) -> int:
        return int(self.lib.pf_workspace_bytes(self._h, batch))

    def set_defer_params(self, on: bool):
        """Deferred ParamNet branch (pf_set_defer_params): the camera-parameter tensor of a forward is complete in stream order once the NEXT forward has been issued
        or after join_params().  For loops that issue forward after forward (bench.py, inference_stream); inference / inference_batch leave it off."""
        _check(self.lib.pf_set_defer_params(self._h, 1 if on else 0), self._h, "pf_set_defer_params")
        self.defer_params = bool(on)
        if not on:
            self.join_params()

    def join_params(self):
        """Put the current stream behind a pending deferred ParamNet branch (no-op when none is pending)."""
        _check(self.lib.pf_join_params(self._h, _stream_ptr()), self._h, "pf_join_params")
        self._deferred = []

    def params_ready_event(self, stream):
        """With the deferred branch on: an event that completes when the camera parameters of the forward issued LAST are complete.  `stream` (a torch stream of its own,
        not the compute stream) is put behind the branch and the event is recorded there -- the compute stream is not touched, so a host thread can wait for the
        parameters of batch i while batch i + 1 is still running.  The branch stays pending for the engine (the next forward still joins it in stream order)."""
        import torch

        with torch.cuda.stream(stream):
            _check(self.lib.pf_join_params(self._h, ctypes.c_void_p(stream.cuda_stream)), self._h, "pf_join_params")
            ev = torch.cuda.Event()
            ev.record(stream)
        return ev

    def _order_scratch(self, name: str, buf):
        """Engine-owned scratch (forward workspace, resize / post-process tables) is reused call after call.  Calls on ONE
        stream are ordered by the stream; when the caller's current stream changes (inference_batch on the default stream,
        then inference_stream's compute stream), the new stream is put behind the last user first -- otherwise two forwards
        on two streams would run in the same workspace at once."""
        import torch

        cur = torch.cuda.current_stream(self.device)
        last = self._scratch_stream.get(name)
        if last is not None and last != cur:
            cur.wait_stream(last)
            if buf is not None:
                buf.record_stream(cur)  # allocated on another stream: keep the allocator from recycling it under this one
        self._scratch_stream[name] = cur
        return buf

    def _workspace(self, nbytes: int):
        import torch

        if self._ws is None or self._ws.numel() < nbytes:
            if self._ws is not None and getattr(self, "defer_params", False):
                self.join_params()  # a pending deferred ParamNet branch still works in the buffer that is about to be released: the allocator does not know the engine's stream
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._order_scratch("ws", self._ws)

    def forward(self, images) -> Tuple[object, object, Optional[object]]:
        """images: uint8 (B,320,320,3) BGR or float32 (B,3,320,320) BGR 0..255, on self.device.
        Returns (pred_gravity (B,Cg,320,320), pred_latitude (B,Cl,320,320), params (B,8) or None)."""
        import torch

        if images.device != self.device:
            raise PfError(f"input on {images.device}, engine on {self.device}")
        images = images.contiguous()
        B = images.shape[0]
        if B > self.max_batch:
            raise PfError(f"batch {B} exceeds PF_MAX_BATCH = {self.max_batch} per forward; split it (PerspectiveFields.inference_batch does)")
        if images.dtype == torch.uint8:
            if tuple(images.shape[1:]) != (NET, NET, 3):
                raise PfError(f"uint8 input must be (B,{NET},{NET},3), got {tuple(images.shape)}")
            fn = self.lib.pf_forward_u8
        elif images.dtype == torch.float32:
            if tuple(images.shape[1:]) != (3, NET, NET):
                raise PfError(f"float32 input must be (B,3,{NET},{NET}), got {tuple(images.shape)}")
            fn = self.lib.pf_forward_f32
        else:
            raise PfError(f"unsupported input dtype {images.dtype}")
        if images.dtype == torch.uint8 and B <= self.graph_max_batch and self.lib.pf_is_tuned(self._h, B):
            return self._forward_graph(images)
        with torch.cuda.device(self.device):
            pg = torch.empty((B, self.gravity_channels, NET, NET), dtype=torch.float32, device=self.device)
            pl = torch.empty((B, self.latitude_channels, NET, NET), dtype=torch.float32, device=self.device)
            params = torch.empty((B, PARAMS_STRIDE), dtype=torch.float32, device=self.device) if self.param_outputs else None
            need = self.workspace_bytes(B)
            ws = self._workspace(need)
            if images.dtype == torch.uint8 and not self.lib.pf_is_tuned(self._h, B):
                fn = self.lib.pf_autotune  # first call with this batch size: same forward, plus per-shape tile timing
            rc = fn(
                self._h, B, images.data_ptr(), pg.data_ptr(), pl.data_ptr(),
                params.data_ptr() if params is not None else None, ws.data_ptr(), ws.numel(), _stream_ptr(),
            )
        _check(rc, self._h, "pf_forward")
        if getattr(self, "defer_params", False) and params is not None:
            # The branch writes `params` on the engine's own stream, which the caching allocator knows nothing about: if the caller dropped its reference, the block could be
            # handed out again on the caller's stream while the branch is still writing.  The branch of forward i is joined by forward i + 1 (in stream order, in front of
            # its decoders), so the tensor of forward i must stay alive until forward i + 1 HAS BEEN ISSUED: keep the last two.
            self._deferred = self._deferred[-1:] + [params]
        return pg, pl, params

    def forward_debug(self, images, shadow=True, ranges=True):
        """pf_debug_forward_u8: the forward plus shadow taps and / or range records (include/pf_hip.h).  images: (B,320,320,3) uint8 on the device.
        Returns (pred_gravity, pred_latitude, params, taps, ranges): taps = {name: NHWC fp32 tensor} in forward order (every MiT / ConvNeXt block output, the stage
        outputs c1..c4, ll, the decoders' conv0 maps, the ParamNet input and stem); ranges = list of dicts (name, elems, max_abs, rms, saturated, non_finite) for every
        tensor that enters a dense contraction."""
        import collections

        import torch

        images = images.contiguous()
        B = images.shape[0]
        with torch.cuda.device(self.device):
            pg = torch.empty((B, self.gravity_channels, NET, NET), dtype=torch.float32, device=self.device)
            pl = torch.empty((B, self.latitude_channels, NET, NET), dtype=torch.float32, device=self.device)
            params = torch.empty((B, PARAMS_STRIDE), dtype=torch.float32, device=self.device) if self.param_outputs else None
            ws = self._workspace(self.workspace_bytes(B))
            nb = int(self.lib.pf_debug_tap_bytes(self._h, B)) if shadow else 0
            tapbuf = torch.empty(max(nb, 256), dtype=torch.uint8, device=self.device)
            rc = self.lib.pf_debug_forward_u8(self._h, B, images.data_ptr(), pg.data_ptr(), pl.data_ptr(), params.data_ptr() if params is not None else None,
                                              ws.data_ptr(), ws.numel(), (1 if shadow else 0) | (2 if ranges else 0), tapbuf.data_ptr(), tapbuf.numel(), _stream_ptr())
        _check(rc, self._h, "pf_debug_forward_u8")
        taps = collections.OrderedDict()
        n = self.lib.pf_debug_taps(self._h, 0, None, None, None)
        if n > 0:
            names = ctypes.create_string_buffer(64 * n)
            offs = (ctypes.c_longlong * n)()
            shp = (ctypes.c_int * (4 * n))()
            self.lib.pf_debug_taps(self._h, n, names, offs, shp)
            for i in range(n):
                nm = names.raw[64 * i:64 * i + 64].split(b"\0")[0].decode()
                sh = tuple(shp[4 * i:4 * i + 4])
                cnt = sh[0] * sh[1] * sh[2] * sh[3]
                taps[nm] = tapbuf[offs[i]:offs[i] + 4 * cnt].view(torch.float32).view(sh)
        out = []
        n = self.lib.pf_debug_ranges(self._h, 0, None, None, None)
        if n > 0:
            names = ctypes.create_string_buffer(96 * n)
            el = (ctypes.c_longlong * n)()
            st = (ctypes.c_float * (4 * n))()
            _check(min(0, self.lib.pf_debug_ranges(self._h, n, names, el, st)), self._h, "pf_debug_ranges")
            for i in range(n):
                nm = names.raw[96 * i:96 * i + 96].split(b"\0")[0].decode()
                out.append({"name": nm, "elems": int(el[i]), "max_abs": float(st[4 * i]), "rms": float((st[4 * i + 1] / max(el[i], 1)) ** 0.5),
                            "saturated": int(st[4 * i + 2]), "non_finite": int(st[4 * i + 3])})
        return pg, pl, params, taps, out

    def _forward_graph(self, images):
        """Small batches: persistent input / output / workspace buffers + a hipGraph captured on first use
        (pf_forward_u8_graph); the results are cloned, so the caller owns them as with forward()."""
        import torch

        B = images.shape[0]
        bufs = self._graph_bufs.get(B)
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_graph_stream", None) is None:
            self._graph_stream = torch.cuda.Stream(device=self.device)  # the legacy default stream cannot be captured
        gs = self._graph_stream
        with torch.cuda.device(self.device):
            if bufs is None:
                bufs = {
                    "in": torch.empty((B, NET, NET, 3), dtype=torch.uint8, device=self.device),
                    "pg": torch.empty((B, self.gravity_channels, NET, NET), dtype=torch.float32, device=self.device),
                    "pl": torch.empty((B, self.latitude_channels, NET, NET), dtype=torch.float32, device=self.device),
                    "params": torch.empty((B, PARAMS_STRIDE), dtype=torch.float32, device=self.device) if self.param_outputs else None,
                    "ws": torch.empty(self.workspace_bytes(B), dtype=torch.uint8, device=self.device),
                }
                self._graph_bufs[B] = bufs
            cur.wait_stream(gs)  # the previous replay (possibly issued from another stream) has finished reading the input buffer
            bufs["in"].copy_(images)
            p = bufs["params"]
            if bufs.get("done") is not None:
                gs.wait_event(bufs["done"])  # the previous caller's clones of the persistent outputs
            gs.wait_stream(cur)  # the input copy
            rc = self.lib.pf_forward_u8_graph(self._h, B, bufs["in"].data_ptr(), bufs["pg"].data_ptr(), bufs["pl"].data_ptr(),
                                              p.data_ptr() if p is not None else None, bufs["ws"].data_ptr(), bufs["ws"].numel(),
                                              ctypes.c_void_p(gs.cuda_stream))
            _check(rc, self._h, "pf_forward_u8_graph")
            cur.wait_stream(gs)
            out = bufs["pg"].clone(), bufs["pl"].clone(), (p.clone() if p is not None else None)
            bufs["done"] = torch.cuda.Event()
            bufs["done"].record(cur)
            return out

    def resize_into(self, img_u8, out_u8_320):
        """Bit-exact PIL BILINEAR resize on the device: img_u8 (H,W,3) uint8 cuda -> out_u8_320 (320,320,3) uint8 cuda view."""
        import torch

        H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
        need = int(self.lib.pf_resize_workspace_bytes(H, W))
        if getattr(self, "_rs_ws", None) is None or self._rs_ws.numel() < need:
            self._rs_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._order_scratch("rs", self._rs_ws)
        with torch.cuda.device(self.device):
            rc = self.lib.pf_resize_bilinear_u8(self._h, img_u8.data_ptr(), H, W, out_u8_320.data_ptr(),
                                                self._rs_ws.data_ptr(), self._rs_ws.numel(), _stream_ptr())
        _check(rc, self._h, "pf_resize_bilinear_u8")

    def resize_batch_into(self, imgs_u8, out_u8):
        """The same for a list of (H_i,W_i,3) uint8 cuda tensors in two launches per 32 images: out_u8 (B,320,320,3) uint8 cuda."""
        import torch

        B = len(imgs_u8)
        hw = [(int(t.shape[0]), int(t.shape[1])) for t in imgs_u8]
        need = 256 + sum((h * NET * 3 + 255) // 256 * 256 for h, _ in hw)
        if getattr(self, "_rs_ws", None) is None or self._rs_ws.numel() < need:
            self._rs_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._order_scratch("rs", self._rs_ws)
        ptrs = (ctypes.c_void_p * B)(*[t.data_ptr() for t in imgs_u8])
        hw_c = (ctypes.c_int32 * (2 * B))(*[v for s in hw for v in s])
        with torch.cuda.device(self.device):
            rc = self.lib.pf_resize_batch_u8(self._h, B, ptrs, hw_c, out_u8.data_ptr(), self._rs_ws.data_ptr(), self._rs_ws.numel(), _stream_ptr())
        _check(rc, self._h, "pf_resize_batch_u8")

    PROFILE_CLASSES = ("igemm", "attention", "layernorm", "dwconv3x3_gelu", "dwconv7x7", "upsample2x", "other", "igemm_sb")

    def profile_begin(self, classes=None, large_only=False):
        """large_only: bracket only the launches of >= 200 GFLOP (PF_PROFILE_LARGE_ONLY): negligible overhead, side stream stays on."""
        mask = 0x80000000 if large_only else 0
        for i, n in enumerate(self.PROFILE_CLASSES):
            if classes is None or n in classes:
                mask |= 1 << i
        _check(self.lib.pf_profile_begin(self._h, mask), self._h, "pf_profile_begin")

    def profile_pause(self):
        """Stop bracketing launches; no synchronisation (records are read by profile_end)."""
        _check(self.lib.pf_profile_pause(self._h), self._h, "pf_profile_pause")

    def profile_end(self) -> Dict[str, dict]:
        n = len(self.PROFILE_CLASSES)
        ms, work, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_long * n)()
        _check(self.lib.pf_profile_end(self._h, ms, work, cnt, n), self._h, "pf_profile_end")
        return {name: {"ms": ms[i], "work": work[i], "launches": cnt[i]} for i, name in enumerate(self.PROFILE_CLASSES)}

    def profile_records(self):
        """Per-launch records of the last profile window: list of (class, work, ms, (M, N, K, KH))."""
        n = self.lib.pf_profile_records(self._h, 0, None, None, None, None)
        cat, work, ms, mnk = (ctypes.c_int * n)(), (ctypes.c_double * n)(), (ctypes.c_float * n)(), (ctypes.c_int * (4 * n))()
        self.lib.pf_profile_records(self._h, n, cat, work, ms, mnk)
        return [(self.PROFILE_CLASSES[cat[i]], work[i], ms[i], tuple(mnk[4 * i : 4 * i + 4])) for i in range(n)]

    PHASES = ("backbone", "low_level_encoder", "decoders", "paramnet", "postprocess")

    def profile_phases(self) -> Dict[str, float]:
        """Component split of the last FULL profile window (pf_profile_phases; call after profile_end): elapsed stream ms per component of the path."""
        n = len(self.PHASES)
        ms = (ctypes.c_double * n)()
        _check(min(0, self.lib.pf_profile_phases(self._h, n, ms)), self._h, "pf_profile_phases")
        return {name: ms[i] for i, name in enumerate(self.PHASES)}

    def postprocess_batch(self, pred_gravity, pred_latitude, sizes):
        """Whole batch in one launch: (B,Cg,320,320), (B,Cl,320,320), [(H, W)] * B -> list of ((2,H,W) unit up-vectors,
        (H,W) degrees).  The outputs of one call are views of one device allocation."""
        import torch

        B = len(sizes)
        pg, pl = pred_gravity.contiguous(), pred_latitude.contiguous()
        counts = [3 * int(h) * int(w) for h, w in sizes]
        with torch.cuda.device(self.device):
            flat = torch.empty(sum(counts), dtype=torch.float32, device=self.device)
            outs, o = [], 0
            for (h, w), n in zip(sizes, counts):
                h, w = int(h), int(w)
                outs.append((flat[o:o + 2 * h * w].view(2, h, w), flat[o + 2 * h * w:o + n].view(h, w)))
                o += n
            hw = (ctypes.c_int32 * (2 * B))(*[int(v) for s in sizes for v in s])
            ups = (ctypes.c_void_p * B)(*[u.data_ptr() for u, _ in outs])
            lats = (ctypes.c_void_p * B)(*[l.data_ptr() for _, l in outs])
            ws_ptr, ws_n = None, 0
            if self.gravity_channels > 2:
                need = B * 3 * NET * NET * 4 + 256
                if getattr(self, "_pp_ws", None) is None or self._pp_ws.numel() < need:
                    self._pp_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                self._order_scratch("pp", self._pp_ws)
                ws_ptr, ws_n = self._pp_ws.data_ptr(), self._pp_ws.numel()
            rc = self.lib.pf_postprocess_batch(self._h, B, pg.data_ptr(), pl.data_ptr(), hw, ups, lats, ws_ptr, ws_n, _stream_ptr())
        _check(rc, self._h, "pf_postprocess_batch")
        return outs

    def postprocess(self, pred_gravity_i, pred_latitude_i, height: int, width: int):
        """One image: (Cg,320,320), (Cl,320,320) -> (2,H,W) unit up-vectors, (H,W) degrees."""
        import torch

        with torch.cuda.device(self.device):
            up = torch.empty((2, height, width), dtype=torch.float32, device=self.device)
            lat = torch.empty((height, width), dtype=torch.float32, device=self.device)
            ws_ptr, ws_n = None, 0
            if self.gravity_channels != 2:
                ws = self._workspace(max(3 * NET * NET * 4 + 512, self._ws.numel() if self._ws is not None else 0))
                ws_ptr, ws_n = ws.data_ptr(), ws.numel()
            rc = self.lib.pf_postprocess(
                self._h, pred_gravity_i.data_ptr(), pred_latitude_i.data_ptr(), int(height), int(width),
                up.data_ptr(), lat.data_ptr(), ws_ptr, ws_n, _stream_ptr(),
            )
        _check(rc, self._h, "pf_postprocess")
        return up, lat
