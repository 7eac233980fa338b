"""GPU parity of the 1-D x 2-D libfrencutils variants (create_xgrid_1dx2d_order1/2, create_xgrid_2dx1d_order1/2, clip,
box_ctrlon/ctrlat, get_grid_area_no_adjust -- SURVEY §8b, B1) against box_oracle.c (pinned bit for bit to the compiled
reference): exchange-cell lists identical, areas / line integrals within 1e-10."""
import ctypes as C

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
RTOL = 1e-10
D2R = np.pi / 180


def _cases(fg):
    c16 = fg.gnomonic_ed_corners(16)
    tl, ta = fg.tripolar_corners(40, 24)
    return {
        "global_box_vs_cubed_equatorial": (np.linspace(0, 360, 37) * D2R, np.linspace(-90, 90, 19) * D2R, 16, 16, c16[0][0], c16[1][0]),
        "global_box_vs_cubed_polar": (np.linspace(0, 360, 37) * D2R, np.linspace(-90, 90, 19) * D2R, 16, 16, c16[0][2], c16[1][2]),
        "shifted_box_vs_tripolar": (np.linspace(-180, 180, 31) * D2R, np.linspace(-80, 88, 22) * D2R, 40, 24, tl, ta),
        "regional_box_vs_latlon": (np.linspace(20, 80, 13) * D2R, np.linspace(-30, 40, 15) * D2R, 36, 18) + fg.latlon_corners(36, 18),
        "single_column_box": (np.array([0.0, 360.0]) * D2R, np.linspace(-90, 90, 10) * D2R, 16, 16, c16[0][1], c16[1][1]),
        "fine_box_vs_cubed": (np.linspace(0, 360, 181) * D2R, np.linspace(-90, 90, 91) * D2R, 16, 16, c16[0][4], c16[1][4]),
    }


@pytest.mark.parametrize("name", ["global_box_vs_cubed_equatorial", "global_box_vs_cubed_polar", "shifted_box_vs_tripolar",
                                  "regional_box_vs_latlon", "single_column_box", "fine_box_vs_cubed"])
@pytest.mark.parametrize("box_is_src,order", [(True, 1), (True, 2), (False, 1), (False, 2)])
def test_box_variants_vs_oracle(fg, gpu_ok, name, box_is_src, order):
    lon_b, lat_b, nxq, nyq, lon_q, lat_q = _cases(fg)[name]
    rng = np.random.default_rng(1)
    nm = (lon_b.size - 1) * (lat_b.size - 1) if box_is_src else nxq * nyq
    mask = (rng.uniform(size=nm) > 0.15).astype(np.float64)
    o = orc.orc_create_xgrid_box(box_is_src, order, lon_b, lat_b, nxq, nyq, lon_q, lat_q, mask)
    r = fg.create_xgrid_box(box_is_src, order, lon_b, lat_b, nxq, nyq, lon_q, lat_q, mask)
    assert r[0] == o["n"] > 0
    for got, k in zip(r[1:5], ("i_in", "j_in", "i_out", "j_out")):
        assert np.array_equal(got, o[k]), k
    assert np.max(np.abs(r[5] - o["area"]) / o["area"]) < RTOL
    if order == 2:
        for got, k in ((r[6], "clon"), (r[7], "clat")):
            assert np.max(np.abs(got - o[k])) < RTOL * np.max(np.abs(o[k])), k


def test_box_primitives_vs_oracle(fg, gpu_ok):
    O, L = orc.oracle(), fg.lib()
    rng = np.random.default_rng(3)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for _ in range(40):
        n = int(rng.integers(3, 7))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        x = np.zeros(50); y = np.zeros(50)
        x[:n] = 1.0 + 0.4 * np.cos(ang); y[:n] = 0.3 + 0.4 * np.sin(ang)
        box = [float(v) for v in (0.8 + rng.uniform(-.2, .2), 0.1 + rng.uniform(-.2, .2), 1.3 + rng.uniform(-.2, .2), 0.6 + rng.uniform(-.2, .2))]
        xo1, yo1, xo2, yo2 = (np.zeros(50) for _ in range(4))
        n1 = O.orc_clip(dp(x), dp(y), n, *box, dp(xo1), dp(yo1))
        n2 = L.clip(dp(x), dp(y), n, *box, dp(xo2), dp(yo2))
        assert n1 == n2
        assert np.array_equal(xo1[:n1].view(np.uint64), xo2[:n2].view(np.uint64))       # pure add/mul/div: bit-identical
        assert np.array_equal(yo1[:n1].view(np.uint64), yo2[:n2].view(np.uint64))
        clon = float(rng.uniform(0, 6))
        a, b = O.orc_box_ctrlat(*box), L.box_ctrlat(*box)
        assert abs(a - b) <= RTOL * abs(a)
        a, b = O.orc_box_ctrlon(*box, clon), L.box_ctrlon(*box, clon)
        assert abs(a - b) <= RTOL * max(abs(a), 1.0)
    lo, la = fg.latlon_corners(12, 9, -30.0, 90.0, -60.0, 70.0)
    a1, a2 = np.zeros(108), np.zeros(108)
    lo1, la1 = np.ascontiguousarray(lo).ravel(), np.ascontiguousarray(la).ravel()
    O.orc_get_grid_area_no_adjust(12, 9, dp(lo1), dp(la1), dp(a1))
    L.get_grid_area_no_adjust(C.byref(C.c_int(12)), C.byref(C.c_int(9)), dp(lo1), dp(la1), dp(a2))
    assert np.max(np.abs(a1 - a2) / np.abs(a1)) < RTOL
