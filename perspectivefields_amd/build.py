"""Builds libpf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# The product library lives in lib/.  The tuning build (PF_TUNING_BUILD=1: ablation kernels, rejected variants) gets its own directory, lib_tune/, so that both can
# be prebuilt in the tree and travel to the GPU box together.
# PF_LIB_SUFFIX=<s> (with PF_SKIP_DIGEST_CHECK=1) selects lib<s>/: a library built from OTHER sources kept next to the product for a same-box A/B.
# PF_ASAN=1: lib_asan/ -- the HOST code of every translation unit under AddressSanitizer + UndefinedBehaviorSanitizer (device code unchanged), for the host-side
# orchestration of engine.hip (stack allocator, checkpoint / tile-table parsers, weight repack): driven on CPU through pf_create(PF_DEVICE_NONE) by tests/test_host_asan.py
ASAN = os.environ.get("PF_ASAN", "0") == "1"
_VARIANT = ("_tune" if os.environ.get("PF_TUNING_BUILD", "0") == "1" else "") + ("_asan" if ASAN else "") + os.environ.get("PF_LIB_SUFFIX", "")
LIBDIR = os.path.join(HERE, "lib" + _VARIANT)
LIB = os.path.join(LIBDIR, "libpf_hip.so")
SOURCES = ["igemm.hip", "igemm_sb.hip", "igemm_sbf.hip", "igemm_sbh.hip", "wino.hip", "attn.hip", "attn_block.hip", "stem7.hip", "thin_linear.hip", "elem.hip", "dw7.hip", "dw7_pk.hip", "cnx_mlp.hip", "mit_mlp.hip", "rb_gemm.hip", "rb_chain.hip", "engine.hip", "engine_ops.hip"]
# dw7.hip: the scalar one-channel-per-lane kernel must not be SLP-vectorised (see the file header)
# NO_PK_F32_FLAGS: these units are compiled without the packed-fp32 feature (pf_kernels.h PF_NO_PK_F32 says why); the x86 host pass prints an "ignoring feature"
# line per function for it, filtered below
NO_PK_F32_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EXTRA_FLAGS = {"dw7.hip": ["-fno-slp-vectorize"], "dw7_pk.hip": ["-fno-slp-vectorize"],
               "attn_block.hip": NO_PK_F32_FLAGS, "stem7.hip": NO_PK_F32_FLAGS, "thin_linear.hip": NO_PK_F32_FLAGS, "cnx_mlp.hip": NO_PK_F32_FLAGS, "mit_mlp.hip": NO_PK_F32_FLAGS, "rb_gemm.hip": NO_PK_F32_FLAGS, "rb_chain.hip": NO_PK_F32_FLAGS}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]
# PF_TUNING_BUILD=1: also compile the measured-and-rejected kernel variants and the ablation (no-load / no-store) kernels that
# the scripts under scripts/ can select; the product build carries the default path and its parity alternatives only
if os.environ.get("PF_TUNING_BUILD", "0") == "1":
    FLAGS.append("-DPF_TUNING_BUILD")
ASAN_HOST_FLAGS = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=undefined"]
if ASAN:
    FLAGS = FLAGS + ASAN_HOST_FLAGS


def asan_runtime() -> str:
    """the shared ASan runtime of hipcc's clang: LD_PRELOAD it into the python process that loads lib_asan/libpf_hip.so"""
    clang = os.path.join(os.path.dirname(os.path.realpath(_hipcc())), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    return subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libpf_hip.so)")


def _digest() -> str:
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".h", ".cpp")):
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(f.encode())
                    h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libpf_hip.digest")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    # objects of translation units that no longer exist (a deleted or renamed .hip) must not linger next to the library that travels to the GPU box
    keep = {src.replace(".hip", ".o") for src in SOURCES}
    for f in os.listdir(LIBDIR):
        if f.endswith(".o") and f not in keep:
            os.remove(os.path.join(LIBDIR, f))
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        extra = list(EXTRA_FLAGS.get(src, []))
        if src == "engine.hip":
            extra.append(f'-DPF_BUILD_DIGEST="{dig}"')  # pf_build_digest(): the loader compares it with the sources it sees
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        out = "\n".join(l for l in out.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}:\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("libpf_hip.so build failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB] + (["-fsanitize=address,undefined", "-shared-libsan"] if ASAN else [])
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
