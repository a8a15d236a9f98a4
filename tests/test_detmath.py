"""csrc/detmath.h (compiled for the host into liboracle_det.so) against glibc libm."""
import numpy as np
import pytest


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a.view(np.int64) - b.view(np.int64)).astype(np.float64)
    d[(np.isnan(a) & np.isnan(b)) | (a == b)] = 0
    return d


@pytest.mark.parametrize("name,lo,hi,tol", [
    ("exp", -50, 50, 2), ("exp", -740, 700, 2), ("sin", -10, 10, 2), ("cos", -10, 10, 2),
    ("sin", -1e5, 1e5, 2), ("tan", -1.5, 1.5, 4), ("tan", -10, 10, 4), ("atan", -5, 5, 2),
    ("atan", -1e4, 1e4, 2)])
def test_detmath_close_to_libm(orc_det, orc_libm, name, lo, hi, tol):
    rng = np.random.default_rng(abs(int(lo * 7)) + len(name))
    x = rng.uniform(lo, hi, 20000)
    d = ulp_diff(orc_det.math(name, x), orc_libm.math(name, x))
    assert d.max() <= tol, (name, d.max())


def test_detmath_hypot(orc_det, orc_libm):
    rng = np.random.default_rng(5)
    x, y = rng.uniform(-300, 300, 20000), rng.uniform(-300, 300, 20000)
    assert ulp_diff(orc_det.math("hypot", x, y), orc_libm.math("hypot", x, y)).max() <= 1


def test_detmath_special_values(orc_det):
    m = orc_det.math
    assert m("exp", [0.0])[0] == 1.0
    assert m("exp", [710.0])[0] == np.inf
    assert m("exp", [-746.0])[0] == 0.0
    assert m("exp", [-745.0])[0] == 5e-324
    assert np.isnan(m("exp", [np.nan])[0])
    assert m("sin", [0.0])[0] == 0.0 and m("cos", [0.0])[0] == 1.0 and m("tan", [0.0])[0] == 0.0
    assert np.isnan(m("sin", [np.inf])[0]) and np.isnan(m("cos", [np.nan])[0]) and np.isnan(m("tan", [-np.inf])[0])
    assert np.isfinite(m("sin", [1e10])[0])  # meaningless beyond 2^30 but defined (and host == device)
    assert m("atan", [np.inf])[0] == np.pi / 2 and m("atan", [-np.inf])[0] == -np.pi / 2
    assert m("atan", [1.0])[0] == np.arctan(1.0)
