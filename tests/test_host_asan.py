"""SURVEY 5 (sanitizer build of the host side): libpf_hip.so with the HOST code of every translation unit under AddressSanitizer + UndefinedBehaviorSanitizer
(PF_ASAN=1 python -m perspectivefields_amd.build -> perspectivefields_amd/lib_asan/), driven without a GPU through pf_create(PF_DEVICE_NONE) by tests/asan_driver.py
in a subprocess that has the ASan runtime preloaded.  Any report (heap / stack overflow, use after free, signed overflow, misaligned access, ...) fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_side_under_asan_ubsan():
    env = dict(os.environ, PF_ASAN="1")
    env.pop("PF_TUNING_BUILD", None)
    b = subprocess.run([sys.executable, "-c", "from perspectivefields_amd import build; print(build.build(verbose=False)); print(build.asan_runtime())"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    if b.returncode != 0:
        pytest.fail("sanitizer build failed:\n" + b.stdout[-2000:] + b.stderr[-4000:])
    rt = b.stdout.strip().splitlines()[-1]
    assert os.path.exists(rt), rt
    env.update(LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert r.returncode == 0 and "ASAN_DRIVER_OK" in r.stdout, out[-6000:]
