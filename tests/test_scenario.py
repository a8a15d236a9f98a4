"""Host-side scenario construction (csrc/scenario.cpp) against the reference's Python spline
(tests/golden/spline_vectors.npz, produced by importing scripts/utils/cubic_spline.py) and against
the facts SURVEY.md records about the reference's sampling."""
import ctypes as C

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def spl():
    return dict(np.load(GOLDEN / "spline_vectors.npz"))


@pytest.mark.parametrize("name", ["two_straight", "two_borrow", "three_straight", "three_bend"])
def test_spline_positions_match_reference_python(pkg, spl, name):
    lib = pkg._lib.load()
    wx, wy = np.ascontiguousarray(spl[name + "_wx"]), np.ascontiguousarray(spl[name + "_wy"])
    out = np.zeros(3)
    for s, pos, yaw in zip(spl[name + "_s"], spl[name + "_pos"], spl[name + "_yaw"]):
        rc = lib.cilqr_reference_line_position(wx.ctypes.data, wy.ctypes.data, len(wx), 0.0, float(s), out.ctypes.data)
        assert rc == 0
        np.testing.assert_allclose(out[:2], pos, rtol=0, atol=1e-10)
        assert abs(out[2] - yaw) < 1e-10
    # lateral offset: (x - w sin(yaw), y + w cos(yaw))  (utils.cpp:28-29)
    s = float(spl[name + "_s"][57])
    lib.cilqr_reference_line_position(wx.ctypes.data, wy.ctypes.data, len(wx), 3.6, s, out.ctypes.data)
    pos, yaw = spl[name + "_pos"][57], spl[name + "_yaw"][57]
    np.testing.assert_allclose(out[:2], [pos[0] - 3.6 * np.sin(yaw), pos[1] + 3.6 * np.cos(yaw)], atol=1e-10)


def test_reference_line_sampling(pkg, scenarios):
    """sample counts recorded by the survey's independent probe: L = 2101 / 1715."""
    _, two = scenarios["two_straight"]
    _, bend = scenarios["three_bend"]
    assert two.lane.size() == 2101 and bend.lane.size() == 1715
    # accumulating s += 0.1 (utils.cpp:25)
    s = 0.0
    acc = []
    while s <= 210.0 and len(acc) < 5000:
        acc.append(s)
        s += 0.1
    assert len(acc) == 2101
    np.testing.assert_array_equal(two.lane.longitude, np.array(acc))
    np.testing.assert_allclose(two.lane.x, -10 + np.array(acc), atol=1e-9)
    np.testing.assert_allclose(two.lane.y, 0.0, atol=1e-12)
    assert two.road_borders.tolist() == [5.4, -1.8] and bend.road_borders.tolist() == [9.0, -1.8]
    assert len(two.center_lines) == 2 and len(two.borders) == 3
    np.testing.assert_allclose(two.center_lines[1].y, 3.6, atol=1e-12)


def test_routes(pkg, scenarios):
    """motion_planning.cpp:121-173 without noise: constant speed along the nearest centre line."""
    cfg, two = scenarios["two_straight"]
    assert two.routes.shape == (4, 220, 3)  # t = 0; t < 12 + 10; t += 0.1
    assert two.line_num.tolist() == [0, 0, 1, 1]
    ic = two.initial_conditions
    t = np.cumsum(np.r_[0.0, np.full(219, 0.1)])
    for v in range(4):
        np.testing.assert_allclose(two.routes[v, :, 0], ic[v, 0] + ic[v, 2] * t, atol=1e-6)
        np.testing.assert_allclose(two.routes[v, :, 1], ic[v, 1], atol=1e-9)
    # opposite-direction vehicles: s decreases, yaw = fmod(yaw + pi, 2 pi)  (motion_planning.cpp:152-158)
    _, bor = scenarios["two_borrow"]
    v = 3
    assert bor.initial_conditions[v, 3] > np.pi / 2
    assert bor.routes[v, 10, 0] < bor.routes[v, 0, 0]
    np.testing.assert_allclose(bor.routes[v, :, 2], np.pi, atol=1e-9)
    _, bend = scenarios["three_bend"]
    assert bend.routes.shape == (4, 250, 3) and bend.line_num.tolist() == [0, 0, 1, 2]


def test_bad_arguments(pkg):
    lib = pkg._lib.load()
    cnt = C.c_int32(0)
    one = np.zeros(1)
    assert lib.cilqr_reference_line_build(one.ctypes.data, one.ctypes.data, 1, 0.0, 0.1, None, None, None, None, 0, C.byref(cnt)) == -1
    dec = np.array([0.0, 1.0]), np.array([0.0, 0.0])
    out = np.zeros(3)
    assert lib.cilqr_reference_line_position(dec[0].ctypes.data, dec[1].ctypes.data, 2, 0.0, 5.0, out.ctypes.data) == -1
