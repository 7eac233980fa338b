"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): exchange-cell counts and indices bit-exact; areas, clon/clat and
weights within 1e-10 relative; remapped fields within 1e-6 relative (we get them bit-exact for equal weights).
"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def relerr(a, b):
    scale = np.maximum(np.abs(b), 1e-10 * np.max(np.abs(b)) + 1e-300)
    return float(np.max(np.abs(a - b) / scale)) if len(b) else 0.0


@pytest.fixture(scope="module")
def c48(fg):
    return fg.gnomonic_ed_corners(48)


def test_get_grid_area(fg, gpu_ok, c48):
    lon, lat = c48
    for t in (0, 2):
        a = fg.get_grid_area(48, 48, lon[t], lat[t])
        b = orc.orc_get_grid_area(48, 48, lon[t], lat[t])
        assert relerr(a, b) < RTOL
    lo, la = fg.latlon_corners(180, 90)
    assert relerr(fg.get_grid_area(180, 90, lo, la), orc.orc_get_grid_area(180, 90, lo, la)) < RTOL


def test_cell_struct_bitwise(fg, gpu_ok, c48):
    """get_grid_cell_struct semantics (create_xgrid.c:991-1016): everything but the area is pure
    IEEE arithmetic and must match the oracle bit for bit, pole cells included."""
    lon, lat = c48
    lo, la = fg.latlon_corners(144, 90)
    g_in = [fg.GridConfig(48, 48, lon[2], lat[2])]
    plan = fg.XgridPlan.create(1, g_in, fg.GridConfig(144, 90, lo, la))
    for which, (nx, ny, x, y) in enumerate([(48, 48, lon[2], lat[2]), (144, 90, lo, la)]):
        d = plan.get_cell_struct(which, nx * ny)
        o = orc.orc_cell_struct(nx, ny, x, y)
        for k in ("lat_min", "lat_max", "lon_min", "lon_max", "lon_avg", "vlon", "vlat"):
            assert np.array_equal(d[k].view(np.uint64), o[k].view(np.uint64)), (which, k)
        assert np.array_equal(d["nvert"], o["nvert"])
    assert o["nvert"].max() == 4 and orc.orc_cell_struct(48, 48, lon[2], lat[2])["nvert"].max() == 5
    plan.destroy()


@pytest.mark.parametrize("order,nlon,nlat,tile,nexp", [
    (1, 180, 90, 0, 8460), (1, 180, 90, 2, 14956), (2, 144, 90, 0, 7584), (2, 144, 90, 2, 12784),
    (2, 144, 90, 5, 12784), (2, 360, 180, 3, None)])
def test_create_xgrid_c48(fg, gpu_ok, c48, order, nlon, nlat, tile, nexp):
    lon, lat = c48
    lo, la = fg.latlon_corners(nlon, nlat)
    f = fg.create_xgrid_2dx2d_order1 if order == 1 else fg.create_xgrid_2dx2d_order2
    r = f(48, 48, nlon, nlat, lon[tile], lat[tile], lo, la)
    o = orc.orc_create_xgrid(order, 48, 48, nlon, nlat, lon[tile], lat[tile], lo, la)
    assert r[0] == o["n"]
    if nexp is not None:
        assert r[0] == nexp
    for a, k in zip(r[1:5], ("i_in", "j_in", "i_out", "j_out")):
        assert np.array_equal(a, o[k]), k
    assert relerr(r[5], o["area"]) < RTOL
    if order == 2:
        # clon/clat are integrals that can cancel to ~0: compare against the area scale
        for a, k in ((r[6], "clon"), (r[7], "clat")):
            assert np.max(np.abs(a - o[k])) <= RTOL * np.max(np.abs(o[k])), k
    # stronger than the bar: bit-identical (libm-sequence sin/cos on the device, csrc/sincos_glibc.h)
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    if orc.host_has_fma():
        assert np.array_equal(bits(r[5]), bits(o["area"]))
        if order == 2:
            assert np.array_equal(bits(r[6]), bits(o["clon"])) and np.array_equal(bits(r[7]), bits(o["clat"]))


def test_device_sincos_equals_host_libm(fg, gpu_ok):
    """The device's latitude trig against the oracle's libm calls on 3 M arguments: plain sin/cos bit for bit."""
    import ctypes as C
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi / 2, np.pi / 2, 2000000), rng.uniform(-0.126, 0.126, 500000),
                        rng.uniform(-1, 1, 500000) * 2.0 ** (-rng.integers(0, 40, 500000))])
    s, c, rs, rc = (np.empty_like(x) for _ in range(4))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert fg.lib().fg_sincos_batch(x.size, dp(x), dp(s), dp(c), 0) == 0
    O = orc.oracle()
    O.orc_sincos.argtypes = [C.c_long] + [C.POINTER(C.c_double)] * 3
    O.orc_sincos.restype = None
    O.orc_sincos(x.size, dp(x), dp(rs), dp(rc))
    if orc.host_has_fma():
        assert np.array_equal(bits(s), bits(rs)) and np.array_equal(bits(c), bits(rc))
    else:
        assert np.mean(bits(s) != bits(rs)) < 2e-3 and np.max(np.abs(s - rs)) < 3e-16


def test_create_xgrid_masked(fg, gpu_ok, c48):
    lon, lat = c48
    lo, la = fg.latlon_corners(90, 45)
    mask = ((np.arange(48 * 48) % 7) != 0).astype(np.float64)
    r = fg.create_xgrid_2dx2d_order1(48, 48, 90, 45, lon[1], lat[1], lo, la, mask)
    o = orc.orc_create_xgrid(1, 48, 48, 90, 45, lon[1], lat[1], lo, la, mask)
    assert r[0] == o["n"]
    for a, k in zip(r[1:5], ("i_in", "j_in", "i_out", "j_out")):
        assert np.array_equal(a, o[k]), k
    assert relerr(r[5], o["area"]) < RTOL
