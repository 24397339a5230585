"""Parity metrics shared by the CPU and GPU tests (tolerances from BASELINE.json:north_star):
up-vector cosine within 1e-3, latitude L1 within 1e-3, ParamNet scalars within 1e-4."""
import numpy as np

TOL_COS = 1e-3      # max over pixels of 1 - cos(angle between up-vectors)
TOL_LAT_L1 = 1e-3   # mean |delta| of latitude (sin units at 320^2, degrees after post-process)
TOL_PARAM = 1e-4    # |delta| of each ParamNet scalar (degrees / relative focal)


def one_minus_cos(a, b):
    """a, b: (2, H, W) unit(ish) vector fields -> per-pixel 1 - cosine."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    num = (a * b).sum(0)
    den = np.sqrt((a * a).sum(0) * (b * b).sum(0))
    ok = den > 0
    out = np.zeros(num.shape)
    out[ok] = 1.0 - num[ok] / den[ok]
    return out


def l1(a, b):
    return float(np.mean(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def assert_fields_close(g, g_ref, lat, lat_ref, what=""):
    c = one_minus_cos(g, g_ref)
    assert c.max() <= TOL_COS, f"{what}: up-vector 1-cos max {c.max():.3e} > {TOL_COS}"
    e = l1(lat, lat_ref)
    assert e <= TOL_LAT_L1, f"{what}: latitude L1 {e:.3e} > {TOL_LAT_L1}"
    return float(c.max()), e
