"""Pin the CPU restatement (oracle/xgrid_oracle.c) against the reference's own code compiled in
place (oracle/_ref): every function must agree BIT FOR BIT.  Runs wherever oracle/_ref exists
(the build container; the GPU box receives the prebuilt .so)."""
import ctypes as C

import numpy as np
import pytest

import gridutil
import orc

pytestmark = pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def fg():
    from conftest import load_package
    return load_package()


@pytest.fixture(scope="module")
def c48(fg):
    return fg.gnomonic_ed_corners(48)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_grid_generator_matches_reference_generator(fg):
    for ni in (8, 48):
        lon_r, lat_r = gridutil.ref_gnomonic_corners(ni)
        lon, lat = fg.gnomonic_ed_corners(ni)
        assert np.array_equal(_bits(lon), _bits(lon_r))
        assert np.array_equal(_bits(lat), _bits(lat_r))
    # tile 1 corner documented at create_xgrid.c:2580
    assert abs(lon[0, 0, 0] * 180 / np.pi - 305.0) < 1e-12
    assert abs(lat[0, 0, 0] * 180 / np.pi + 35.26438968275467) < 1e-12


def test_primitives_bitwise(c48, fg):
    L, R = orc.oracle(), orc.ref()
    lon, lat = c48
    dp = orc.dp
    rng = np.random.default_rng(7)
    cells = [(t, j, i) for t in range(6) for j in (0, 1, 23, 24, 47) for i in (0, 1, 23, 24, 47)]
    for (t, j, i) in cells:
        x = np.array([lon[t, j, i], lon[t, j, i + 1], lon[t, j + 1, i + 1], lon[t, j + 1, i]] + [0.0] * 16)
        y = np.array([lat[t, j, i], lat[t, j, i + 1], lat[t, j + 1, i + 1], lat[t, j + 1, i]] + [0.0] * 16)
        x1, y1, x2, y2 = x.copy(), y.copy(), x.copy(), y.copy()
        n1 = L.orc_fix_lon(x1.ctypes.data_as(dp), y1.ctypes.data_as(dp), 4, np.pi)
        n2 = R.fix_lon(x2.ctypes.data_as(dp), y2.ctypes.data_as(dp), 4, np.pi)
        assert n1 == n2
        assert np.array_equal(_bits(x1[:n1]), _bits(x2[:n1])) and np.array_equal(_bits(y1[:n1]), _bits(y2[:n1]))
        a1 = L.orc_poly_area(x1.ctypes.data_as(dp), y1.ctypes.data_as(dp), n1)
        a2 = R.poly_area(x2.ctypes.data_as(dp), y2.ctypes.data_as(dp), n1)
        assert a1 == a2
        clon = float(np.mean(x1[:n1]))
        assert L.orc_poly_ctrlon(x1.ctypes.data_as(dp), y1.ctypes.data_as(dp), n1, clon) == \
            R.poly_ctrlon(x2.ctypes.data_as(dp), y2.ctypes.data_as(dp), n1, clon)
        assert L.orc_poly_ctrlat(x1.ctypes.data_as(dp), y1.ctypes.data_as(dp), n1) == \
            R.poly_ctrlat(x2.ctypes.data_as(dp), y2.ctypes.data_as(dp), n1)
        # clip against a randomly displaced lat-lon box
        cx, cy = clon + rng.uniform(-0.02, 0.02), float(np.mean(y1[:n1])) + rng.uniform(-0.02, 0.02)
        bx = np.array([cx - 0.02, cx + 0.02, cx + 0.02, cx - 0.02])
        by = np.array([cy - 0.015, cy - 0.015, cy + 0.015, cy + 0.015])
        o1x, o1y, o2x, o2y = (np.zeros(50) for _ in range(4))
        m1 = L.orc_clip_2dx2d(x1.ctypes.data_as(dp), y1.ctypes.data_as(dp), n1, bx.ctypes.data_as(dp), by.ctypes.data_as(dp), 4,
                              o1x.ctypes.data_as(dp), o1y.ctypes.data_as(dp))
        m2 = R.clip_2dx2d(x2.ctypes.data_as(dp), y2.ctypes.data_as(dp), n1, bx.ctypes.data_as(dp), by.ctypes.data_as(dp), 4,
                          o2x.ctypes.data_as(dp), o2y.ctypes.data_as(dp))
        assert m1 == m2
        assert np.array_equal(_bits(o1x[:m1]), _bits(o2x[:m1])) and np.array_equal(_bits(o1y[:m1]), _bits(o2y[:m1]))


def test_get_grid_area_bitwise(c48, fg):
    lon, lat = c48
    for t in (0, 2, 5):
        assert np.array_equal(_bits(orc.orc_get_grid_area(48, 48, lon[t], lat[t])), _bits(orc.ref_get_grid_area(48, 48, lon[t], lat[t])))
    lo, la = fg.latlon_corners(180, 90)
    assert np.array_equal(_bits(orc.orc_get_grid_area(180, 90, lo, la)), _bits(orc.ref_get_grid_area(180, 90, lo, la)))


@pytest.mark.parametrize("order,nlon,nlat,expected", [(1, 180, 90, {0: 8460, 2: 14956}), (2, 144, 90, {0: 7584, 2: 12784})])
def test_create_xgrid_bitwise_c48(c48, fg, order, nlon, nlat, expected):
    """C48 tiles 1 and 3 (polar) against 2-degree / 144x90 targets; counts are BASELINE.md's."""
    lon, lat = c48
    lo, la = fg.latlon_corners(nlon, nlat)
    for t, nexp in expected.items():
        a = orc.orc_create_xgrid(order, 48, 48, nlon, nlat, lon[t], lat[t], lo, la)
        b = orc.ref_create_xgrid(order, 48, 48, nlon, nlat, lon[t], lat[t], lo, la)
        assert a["n"] == b["n"] == nexp
        for k in ("i_in", "j_in", "i_out", "j_out"):
            assert np.array_equal(a[k], b[k]), k
        for k in ("area",) + (("clon", "clat") if order == 2 else ()):
            assert np.array_equal(_bits(a[k]), _bits(b[k])), k
    if order == 1:
        # first exchange cell of tile 1 quoted in BASELINE.md
        a = orc.orc_create_xgrid(1, 48, 48, 180, 90, lon[0], lat[0], lo, la)
        assert (a["i_in"][0], a["j_in"][0], a["i_out"][0], a["j_out"][0]) == (0, 0, 152, 27)
        assert abs(a["area"][0] - 14492669980.22258) < 1e-4


def test_conserve_interp_ref_small(fg):
    """interp.c:262 on a small pair of lat-lon grids: restated arithmetic equals the reference's."""
    R = orc.ref()
    lo1, la1 = fg.latlon_corners(36, 18)
    lo2, la2 = fg.latlon_corners(20, 10)
    rng = np.random.default_rng(3)
    data = rng.standard_normal(36 * 18)
    out = np.empty(200)
    mask = np.ones(36 * 18)
    P = lambda a: np.ascontiguousarray(a).ctypes.data_as(orc.dp)
    R.conserve_interp(36, 18, 20, 10, P(lo1), P(la1), P(lo2), P(la2), P(mask), P(data), P(out))
    x = orc.orc_create_xgrid(1, 36, 18, 20, 10, lo1, la1, lo2, la2)
    dst_area = np.zeros(200)
    for n in range(x["n"]):
        dst_area[x["j_out"][n] * 20 + x["i_out"][n]] += x["area"][n]
    exp = np.zeros(200)
    for n in range(x["n"]):
        d = x["j_out"][n] * 20 + x["i_out"][n]
        exp[d] += data[x["j_in"][n] * 36 + x["i_in"][n]] * (x["area"][n] / dst_area[d])
    assert np.array_equal(_bits(exp), _bits(out))
