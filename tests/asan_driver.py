"""Driver of tests/test_host_asan.py: runs INSIDE a python process that has the ASan runtime preloaded and PF_ASAN=1 set, and walks the host side of the engine
through the C ABI on an engine created with PF_DEVICE_NONE (no GPU): checkpoint loading incl. its error paths, weight finalisation (BN / LayerNorm / layer-scale /
Linear->3x3 folds, conv repack, bf16 / fp16 splits, fragment-order packing of the fused kernels' weights), the workspace dry run of the stack allocator for several
batch sizes and all three architectures, the tile-table writer / parser, and the loud failure of every device entry point."""
import ctypes
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd.config import arch_of, get_cfg  # noqa: E402
from perspectivefields_amd.engine import load_library  # noqa: E402
from perspectivefields_amd.synth import synthetic_state_dict  # noqa: E402

lib = load_library()
assert "lib_asan" in lib._name, lib._name
PF_DEVICE_NONE = -1


def load(h, sd, skip=None, bad_shape=None):
    for key, val in sd.items():
        if key == skip:
            continue
        arr = np.asarray(val)
        if key.endswith("num_batches_tracked"):
            arr = np.zeros((), dtype=np.float32)
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if key == bad_shape:
            arr = np.ascontiguousarray(arr.reshape(-1)[:-1])
        shape = (ctypes.c_int64 * max(arr.ndim, 1))(*arr.shape)
        rc = lib.pf_load_tensor(h, key.encode(), arr.ctypes.data_as(ctypes.c_void_p), shape, arr.ndim)
        if rc != 0:
            return rc
    return 0


for version in ("Paramnet-360Cities-edina-centered", "PersNet-360Cities", "Paramnet-360Cities-edina-uncentered"):
    sd = synthetic_state_dict(version, 0)
    arch = arch_of(get_cfg(version))["arch_id"]
    h = ctypes.c_void_p()
    assert lib.pf_create(ctypes.byref(h), PF_DEVICE_NONE, arch) == 0, lib.pf_last_error(None)
    assert load(h, sd) == 0, lib.pf_last_error(h)
    assert lib.pf_finalize_weights(h) == 0, lib.pf_last_error(h)
    sizes = [int(lib.pf_workspace_bytes(h, b)) for b in (1, 3, 32, 64)]
    assert all(s > 0 for s in sizes) and sizes == sorted(sizes), sizes
    assert int(lib.pf_workspace_bytes(h, 1000)) == 0                    # beyond PF_MAX_BATCH
    # a forward on a host-only engine is a loud device error, not a crash
    buf = (ctypes.c_float * 16)()
    rc = lib.pf_forward_u8(h, 1, ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p),
                           ctypes.cast(buf, ctypes.c_void_p), 1 << 40, None)
    assert rc != 0 and b"PF_DEVICE_NONE" in lib.pf_last_error(h), lib.pf_last_error(h)
    # tile table: write what the engine holds, read it back, read the shipped table, read garbage
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "tiles.txt").encode()
        n_saved = lib.pf_save_tile_table(h, path)
        assert n_saved >= 0
        assert lib.pf_load_tile_table(h, path) == n_saved
        shipped = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "perspectivefields_amd", "tuned", "gfx950_tiles.txt")
        assert lib.pf_load_tile_table(h, shipped.encode()) > 0
        junk = os.path.join(td, "junk.txt")
        open(junk, "w").write("not a tile table\n-1 -1 999999999999 x\n" + "7 " * 500 + "\n")
        lib.pf_load_tile_table(h, junk.encode())                        # must not crash; entries it cannot parse are dropped
        assert lib.pf_load_tile_table(h, os.path.join(td, "missing.txt").encode()) <= 0
    assert lib.pf_destroy(h) == 0
    print(version, "workspace bytes", sizes, flush=True)

# error paths of the loader: a missing tensor, a tensor of the wrong size, an unknown key, finalize twice
version = "Paramnet-360Cities-edina-centered"
sd = synthetic_state_dict(version, 0)
arch = arch_of(get_cfg(version))["arch_id"]
keys = list(sd)
h = ctypes.c_void_p()
assert lib.pf_create(ctypes.byref(h), PF_DEVICE_NONE, arch) == 0
assert load(h, sd, skip=keys[len(keys) // 2]) == 0
assert lib.pf_finalize_weights(h) != 0 and b"missing" in lib.pf_last_error(h)
lib.pf_destroy(h)
h = ctypes.c_void_p()
assert lib.pf_create(ctypes.byref(h), PF_DEVICE_NONE, arch) == 0
rc = load(h, sd, bad_shape=keys[10])
assert rc != 0 or lib.pf_finalize_weights(h) != 0
lib.pf_destroy(h)
h = ctypes.c_void_p()
assert lib.pf_create(ctypes.byref(h), PF_DEVICE_NONE, arch) == 0
assert load(h, sd) == 0
arr = np.zeros((3,), dtype=np.float32)
shape = (ctypes.c_int64 * 1)(3)
lib.pf_load_tensor(h, b"no.such.tensor", arr.ctypes.data_as(ctypes.c_void_p), shape, 1)
assert lib.pf_finalize_weights(h) != 0                                   # strict: an unexpected tensor is fatal
lib.pf_destroy(h)
assert lib.pf_create(ctypes.byref(h), PF_DEVICE_NONE, 99) != 0
print("ASAN_DRIVER_OK", flush=True)
