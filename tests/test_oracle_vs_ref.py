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


GC_CASES = {
    "c24_tile1_144x90": lambda fg: (24, 24, 144, 90) + (fg.gnomonic_ed_corners(24)[0][0], fg.gnomonic_ed_corners(24)[1][0]) + fg.latlon_corners(144, 90),
    "c24_tile3_polar_144x90": lambda fg: (24, 24, 144, 90) + (fg.gnomonic_ed_corners(24)[0][2], fg.gnomonic_ed_corners(24)[1][2]) + fg.latlon_corners(144, 90),
    "latlon_aligned_2x_refinement": lambda fg: (36, 18, 72, 36) + fg.latlon_corners(36, 18) + fg.latlon_corners(72, 36),
    "latlon_regional_offset": lambda fg: (30, 20, 45, 33) + fg.latlon_corners(30, 20, 10., 70., -30., 30.) + fg.latlon_corners(45, 33, 0., 90., -40., 40.),
    "latlon_to_cubed_polar_tile": lambda fg: (40, 20, 12, 12) + fg.latlon_corners(40, 20) + (fg.gnomonic_ed_corners(12)[0][5], fg.gnomonic_ed_corners(12)[1][5]),
    "tripolar_to_cubed": lambda fg: (45, 27, 12, 12) + fg.tripolar_corners(45, 27) + (fg.gnomonic_ed_corners(12)[0][2], fg.gnomonic_ed_corners(12)[1][2]),
}


@pytest.mark.parametrize("name", sorted(GC_CASES))
def test_great_circle_oracle_bitwise(fg, name):
    """gc_oracle.c (array restatement of the Node-list clip) against the reference's own create_xgrid_great_circle /
    get_grid_great_circle_area: identical exchange-cell lists and bit-identical areas, including the degenerate
    aligned-edge cases (u snapped to 0/1, coincident planes) and pole cells (duplicate vertices merged by addEnd)."""
    args = GC_CASES[name](fg)
    o = orc.orc_create_xgrid_gc(*args)
    r = orc.ref_create_xgrid_gc(*args)
    assert o["n"] == r["n"] and o["n"] > 0
    for k in ("i_in", "j_in", "i_out", "j_out"):
        assert np.array_equal(o[k], r[k]), k
    assert np.array_equal(o["area"].view(np.uint64), r["area"].view(np.uint64))
    a1 = orc.orc_get_grid_gc_area(args[0], args[1], args[4], args[5])
    a2 = orc.ref_get_grid_gc_area(args[0], args[1], args[4], args[5])
    assert np.array_equal(a1.view(np.uint64), a2.view(np.uint64))


def test_great_circle_clip_vertices_bitwise(fg):
    """clip_2dx2d_great_circle vertex lists (not just areas) for neighbouring C24 / lat-lon cell pairs."""
    import ctypes as C
    R, O = orc.ref(), orc.oracle()
    lon, lat = fg.gnomonic_ed_corners(24)
    lo, la = fg.latlon_corners(144, 90)
    n1 = 25 * 25
    x1, y1, z1 = (np.empty(n1) for _ in range(3))
    O.orc_latlon2xyz(n1, orc._dp(orc.f64(lon[2]).ravel()), orc._dp(orc.f64(lat[2]).ravel()), orc._dp(x1), orc._dp(y1), orc._dp(z1))
    n2 = 145 * 91
    x2, y2, z2 = (np.empty(n2) for _ in range(3))
    O.orc_latlon2xyz(n2, orc._dp(orc.f64(lo).ravel()), orc._dp(orc.f64(la).ravel()), orc._dp(x2), orc._dp(y2), orc._dp(z2))
    cell = lambda x, nxp, i, j: np.array([x[j * nxp + i], x[(j + 1) * nxp + i], x[(j + 1) * nxp + i + 1], x[j * nxp + i + 1]])
    gc = orc.orc_create_xgrid_gc(24, 24, 144, 90, lon[2], lat[2], lo, la)
    checked = 0
    for k in range(0, gc["n"], 7):
        i1, j1, i2, j2 = (int(gc[key][k]) for key in ("i_in", "j_in", "i_out", "j_out"))
        a = [cell(v, 25, i1, j1) for v in (x1, y1, z1)]
        b = [cell(v, 145, i2, j2) for v in (x2, y2, z2)]
        oo = [np.zeros(50) for _ in range(3)]
        rr = [np.zeros(50) for _ in range(3)]
        no = O.orc_clip_2dx2d_great_circle(*[orc._dp(v) for v in a], 4, *[orc._dp(v) for v in b], 4, *[orc._dp(v) for v in oo])
        nr = R.clip_2dx2d_great_circle(*[orc._dp(v) for v in a], 4, *[orc._dp(v) for v in b], 4, *[orc._dp(v) for v in rr])
        assert no == nr and no >= 3
        for u, v in zip(oo, rr):
            assert np.array_equal(u[:no].view(np.uint64), v[:nr].view(np.uint64))
        checked += 1
    assert checked > 500


def _box_cases(fg):
    D2R = np.pi / 180
    c16 = fg.gnomonic_ed_corners(16)
    tl, ta = fg.tripolar_corners(40, 24)
    return {
        "global_box_vs_cubed_equatorial": (np.linspace(0, 360, 37) * D2R, np.linspace(-90, 90, 19) * D2R, 16, 16, c16[0][0], c16[1][0]),
        "global_box_vs_cubed_polar": (np.linspace(0, 360, 37) * D2R, np.linspace(-90, 90, 19) * D2R, 16, 16, c16[0][2], c16[1][2]),
        "shifted_box_vs_tripolar": (np.linspace(-180, 180, 31) * D2R, np.linspace(-80, 88, 22) * D2R, 40, 24, tl, ta),
        "regional_box_vs_latlon": (np.linspace(20, 80, 13) * D2R, np.linspace(-30, 40, 15) * D2R, 36, 18) + fg.latlon_corners(36, 18),
        "single_column_box": (np.array([0.0, 360.0]) * D2R, np.linspace(-90, 90, 10) * D2R, 16, 16, c16[0][1], c16[1][1]),
    }


@pytest.mark.parametrize("name", ["global_box_vs_cubed_equatorial", "global_box_vs_cubed_polar", "shifted_box_vs_tripolar",
                                  "regional_box_vs_latlon", "single_column_box"])
@pytest.mark.parametrize("box_is_src,order", [(True, 1), (True, 2), (False, 1), (False, 2)])
def test_box_variants_oracle_bitwise(fg, name, box_is_src, order):
    """box_oracle.c against the reference's create_xgrid_1dx2d_order1/2 and create_xgrid_2dx1d_order1/2."""
    lon_b, lat_b, nxq, nyq, lon_q, lat_q = _box_cases(fg)[name]
    rng = np.random.default_rng(1)
    nm = (lon_b.size - 1) * (lat_b.size - 1) if box_is_src else nxq * nyq
    mask = (rng.uniform(size=nm) > 0.15).astype(np.float64)
    o = orc.orc_create_xgrid_box(box_is_src, order, lon_b, lat_b, nxq, nyq, lon_q, lat_q, mask)
    r = orc.ref_create_xgrid_box(box_is_src, order, lon_b, lat_b, nxq, nyq, lon_q, lat_q, mask)
    assert o["n"] == r["n"] and o["n"] > 0
    for k in ("i_in", "j_in", "i_out", "j_out"):
        assert np.array_equal(o[k], r[k]), k
    for k in ("area",) + (("clon", "clat") if order == 2 else ()):
        assert np.array_equal(o[k].view(np.uint64), r[k].view(np.uint64)), k


def test_box_primitives_bitwise(fg):
    R, O = orc.ref(), orc.oracle()
    rng = np.random.default_rng(3)
    for _ in range(300):
        n = int(rng.integers(3, 7))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        x = np.zeros(50); y = np.zeros(50)
        x[:n] = 1.0 + 0.4 * np.cos(ang); y[:n] = 0.3 + 0.4 * np.sin(ang)
        box = [float(v) for v in (0.8 + rng.uniform(-.2, .2), 0.1 + rng.uniform(-.2, .2), 1.3 + rng.uniform(-.2, .2), 0.6 + rng.uniform(-.2, .2))]
        xo1, yo1, xo2, yo2 = (np.zeros(50) for _ in range(4))
        n1 = O.orc_clip(orc._dp(x), orc._dp(y), n, *box, orc._dp(xo1), orc._dp(yo1))
        n2 = R.clip(orc._dp(x), orc._dp(y), n, *box, orc._dp(xo2), orc._dp(yo2))
        assert n1 == n2
        assert np.array_equal(xo1[:n1].view(np.uint64), xo2[:n2].view(np.uint64)) and np.array_equal(yo1[:n1].view(np.uint64), yo2[:n2].view(np.uint64))
        clon = float(rng.uniform(0, 6))
        assert O.orc_box_ctrlat(*box) == R.box_ctrlat(*box)
        assert O.orc_box_ctrlon(*box, clon) == R.box_ctrlon(*box, clon)
    lo, la = fg.latlon_corners(12, 9, -30.0, 90.0, -60.0, 70.0)
    a1, a2 = np.zeros(108), np.zeros(108)
    O.orc_get_grid_area_no_adjust(12, 9, orc._dp(orc.f64(lo).ravel()), orc._dp(orc.f64(la).ravel()), orc._dp(a1))
    R.get_grid_area_no_adjust(C.byref(C.c_int(12)), C.byref(C.c_int(9)), orc._dp(orc.f64(lo).ravel()), orc._dp(orc.f64(la).ravel()), orc._dp(a2))
    assert np.array_equal(a1.view(np.uint64), a2.view(np.uint64))
