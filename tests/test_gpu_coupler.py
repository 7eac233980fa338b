"""fg_plan_get_polygons / fg_plan_create_polylist / coupler.coupler_xgrid (the make_coupler_mosaic consumers).

* The polygons a plan returns are the reference's clip outputs: for every exchange cell of C48 tile 3 -> 2 deg the vertices equal
  what the reference's OWN compiled clip_2dx2d (oracle/_ref) returns for the pair built the way create_xgrid builds it
  (create_xgrid.c:1040-1080), bit for bit; the great-circle plan's against clip_2dx2d_great_circle.
* coupler_xgrid (three device searches) against make_coupler_mosaic.c:1360-1716 restated as a Python loop over the reference's own
  compiled primitives (fix_lon, clip_2dx2d, poly_area, poly_ctrlon, poly_ctrlat): atmosphere x ocean and atmosphere x land lists,
  areas and centroid integrals bit for bit.  The LOOP is a restatement (make_coupler_mosaic.c itself needs netCDF: parity
  unpinned for the driver); every number in it comes out of the reference's object code."""
import ctypes as C

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
DP = C.POINTER(C.c_double)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _cell(lon, lat, nx, i, j):
    """the four corners of cell (i, j) in create_xgrid's order: SW, SE, NE, NW"""
    n0 = j * (nx + 1) + i
    idx = [n0, n0 + 1, n0 + nx + 2, n0 + nx + 1]
    return lon.reshape(-1)[idx].copy(), lat.reshape(-1)[idx].copy()


def _fix_lon(R, x, y, tlon):
    xb = np.zeros(12); yb = np.zeros(12)
    xb[:x.size] = x; yb[:y.size] = y
    n = R.fix_lon(xb.ctypes.data_as(DP), yb.ctypes.data_as(DP), x.size, tlon)
    return xb[:n].copy(), yb[:n].copy()


def _clip(R, x1, y1, x2, y2):
    xo = np.zeros(50); yo = np.zeros(50)
    a, b, c, d = (np.ascontiguousarray(v) for v in (x1, y1, x2, y2))
    n = R.clip_2dx2d(a.ctypes.data_as(DP), b.ctypes.data_as(DP), a.size, c.ctypes.data_as(DP), d.ctypes.data_as(DP), c.size,
                     xo.ctypes.data_as(DP), yo.ctypes.data_as(DP))
    return xo[:n].copy(), yo[:n].copy()


def _poly(R, fn, x, y, *extra):
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    return getattr(R, fn)(x.ctypes.data_as(DP), y.ctypes.data_as(DP), x.size, *extra)


@pytest.mark.skipif(not orc.ref_available(), reason="needs oracle/_ref (the reference compiled in place)")
def test_plan_polygons_are_the_references_clip_outputs(fg, gpu_ok):
    R = orc.ref()
    ni, nlon, nlat = 48, 180, 90
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    t = 2                                                       # the north-polar tile: pole-fixed cells, shifted cells
    for rect in (1, 0):
        fg.lib().fg_set_search_rect(rect)
        try:
            p = fg.XgridPlan.create(2, [fg.GridConfig(ni, ni, lon[t], lat[t])], fg.GridConfig(nlon, nlat, lo, la))
        finally:
            fg.lib().fg_set_search_rect(1)
        x = p.get_xgrid(); poly = p.get_polygons(maxv=16); p.destroy()
        assert x["area"].size == 14956 == poly["n"].size
        rng = np.random.default_rng(1)
        pick = np.unique(np.concatenate([rng.integers(0, x["area"].size, 1500), np.flatnonzero(poly["n"] > 5)[:200], np.arange(200)]))
        for k in pick:
            x1, y1 = _fix_lon(R, *_cell(lon[t], lat[t], ni, x["i_in"][k], x["j_in"][k]), np.pi)          # create_xgrid.c:1040-1048
            x2, y2 = _fix_lon(R, *_cell(lo, la, nlon, x["i_out"][k], x["j_out"][k]), np.pi)              # :1004
            dx = np.sum(x2) / x2.size - np.sum(x1) / x1.size                                            # :1064-1074 (avgval_double)
            if dx < -np.pi:
                x2 = x2 + 2 * np.pi
            elif dx > np.pi:
                x2 = x2 - 2 * np.pi
            xo, yo = _clip(R, x1, y1, x2, y2)
            n = int(poly["n"][k])
            assert n == xo.size, (k, n, xo.size)
            assert np.array_equal(_bits(poly["lon"][k, :n]), _bits(xo)) and np.array_equal(_bits(poly["lat"][k, :n]), _bits(yo)), k
            if orc.host_has_fma():
                assert _bits(np.array([x["area"][k]]))[0] == _bits(np.array([_poly(R, "poly_area", xo, yo)]))[0]


@pytest.mark.skipif(not orc.ref_available(), reason="needs oracle/_ref (the reference compiled in place)")
def test_great_circle_plan_polygons(fg, gpu_ok):
    R = orc.ref()
    ni, nlon, nlat = 12, 36, 18
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    p = fg.XgridPlan.create_great_circle([fg.GridConfig(ni, ni, lon[0], lat[0])], fg.GridConfig(nlon, nlat, lo, la))
    x = p.get_xgrid(); poly = p.get_polygons(maxv=16); p.destroy()
    xs, ys, zs = fg.latlon2xyz(lon[0], lat[0]); xd, yd, zd = fg.latlon2xyz(lo, la)
    for k in range(0, x["area"].size, 3):
        def quad(ax, nx, i, j):                                 # clockwise: create_xgrid.c:1413-1420
            n0 = j * (nx + 1) + i
            return np.ascontiguousarray(ax[[n0, n0 + nx + 1, n0 + nx + 2, n0 + 1]])
        a = [quad(v, ni, x["i_in"][k], x["j_in"][k]) for v in (xs, ys, zs)]
        b = [quad(v, nlon, x["i_out"][k], x["j_out"][k]) for v in (xd, yd, zd)]
        out = [np.zeros(50) for _ in range(3)]
        n = R.clip_2dx2d_great_circle(*[v.ctypes.data_as(DP) for v in a], 4, *[v.ctypes.data_as(DP) for v in b], 4,
                                      *[v.ctypes.data_as(DP) for v in out])
        assert n == poly["n"][k]
        for got, want in zip((poly["x"], poly["y"], poly["z"]), out):
            assert np.array_equal(_bits(got[k, :n]), _bits(want[:n])), k


def _reference_coupler(R, atm, lnd, ocn, omask, order, thresh=1.0e-6, min_frac=1.0e-4):
    """make_coupler_mosaic.c:1360-1716 (legacy clip, no nest, lnd_same_as_atm = 0) over the reference's compiled primitives."""
    area = lambda g: orc.ref_get_grid_area(g.nx, g.ny, g.lonc, g.latc).reshape(-1)
    area_lnd, area_ocn = area(lnd), area(ocn)
    axo = {k: [] for k in ("ta", "ia", "ja", "io", "jo", "area", "clon", "clat")}
    axl = {k: [] for k in ("ta", "ia", "ja", "il", "jl", "area", "clon", "clat")}
    corners = lambda g: (np.asarray(g.lonc).reshape(g.ny + 1, g.nx + 1), np.asarray(g.latc).reshape(g.ny + 1, g.nx + 1))
    (xl_all, yl_all), (xo_all, yo_all) = corners(lnd), corners(ocn)
    cmin = lambda y: np.minimum(np.minimum(y[:-1, :-1], y[:-1, 1:]), np.minimum(y[1:, 1:], y[1:, :-1]))
    cmax = lambda y: np.maximum(np.maximum(y[:-1, :-1], y[:-1, 1:]), np.maximum(y[1:, 1:], y[1:, :-1]))
    yl_min, yl_max, yo_min, yo_max = cmin(yl_all), cmax(yl_all), cmin(yo_all), cmax(yo_all)
    for na, g in enumerate(atm):
        area_atm = area(g)
        for ja in range(g.ny):
            for ia in range(g.nx):
                xa, ya = _cell(np.asarray(g.lonc), np.asarray(g.latc), g.nx, ia, ja)
                ya_min, ya_max = ya.min(), ya.max()
                xa, ya = _fix_lon(R, xa, ya, np.pi)
                xa_min, xa_max, xa_avg = xa.min(), xa.max(), np.sum(xa) / xa.size
                a_area = area_atm[ja * g.nx + ia]
                found = []                                       # the remembered atmosphere x land polygons of this cell
                for jl, il in zip(*np.nonzero(~((yl_min >= ya_max) | (yl_max <= ya_min)))):
                    xl, yl = _fix_lon(R, *_cell(xl_all, yl_all, lnd.nx, il, jl), xa_avg)
                    if xa_min >= xl.max() or xa_max <= xl.min():
                        continue
                    xo, yo = _clip(R, xa, ya, xl, yl)
                    if xo.size and _poly(R, "poly_area", xo, yo) / min(area_lnd[jl * lnd.nx + il], a_area) > thresh:
                        found.append(dict(il=il, jl=jl, x=xo, y=yo, area=0.0, clon=0.0, clat=0.0))
                # (ocean cells whose latitude range misses the atmosphere cell clip to nothing: skipped here, looped over there)
                for jo, io in zip(*np.nonzero(~((yo_min >= ya_max) | (yo_max <= ya_min)))):
                    ocn_frac = omask[jo, io]; lnd_frac = 1 - ocn_frac
                    xo_, yo_ = _fix_lon(R, *_cell(xo_all, yo_all, ocn.nx, io, jo), xa_avg)
                    xo_min, xo_max = xo_.min(), xo_.max()
                    if ocn_frac > min_frac and not (xa_min >= xo_max or xa_max <= xo_min):
                        x_, y_ = _clip(R, xa, ya, xo_, yo_)
                        if x_.size:
                            xarea = _poly(R, "poly_area", x_, y_) * ocn_frac
                            if xarea / min(area_ocn[jo * ocn.nx + io], a_area) > thresh:
                                for k, v in (("ta", na), ("ia", ia), ("ja", ja), ("io", io), ("jo", jo), ("area", xarea)):
                                    axo[k].append(v)
                                axo["clon"].append(_poly(R, "poly_ctrlon", x_, y_, xa_avg) * ocn_frac if order == 2 else 0.0)
                                axo["clat"].append(_poly(R, "poly_ctrlat", x_, y_) * ocn_frac if order == 2 else 0.0)
                    if lnd_frac > min_frac:
                        for f in found:
                            if f["x"].min() >= xo_max or f["x"].max() <= xo_min:
                                continue
                            x_, y_ = _clip(R, f["x"], f["y"], xo_, yo_)
                            if x_.size:
                                xarea = _poly(R, "poly_area", x_, y_) * lnd_frac
                                if xarea / min(area_lnd[f["jl"] * lnd.nx + f["il"]], a_area) > thresh:
                                    f["area"] += xarea
                                    if order == 2:
                                        f["clon"] += _poly(R, "poly_ctrlon", x_, y_, xa_avg) * lnd_frac
                                        f["clat"] += _poly(R, "poly_ctrlat", x_, y_) * lnd_frac
                for f in found:
                    if f["area"] / min(area_lnd[f["jl"] * lnd.nx + f["il"]], a_area) > thresh:
                        for k, v in (("ta", na), ("ia", ia), ("ja", ja), ("il", f["il"]), ("jl", f["jl"]), ("area", f["area"]),
                                     ("clon", f["clon"]), ("clat", f["clat"])):
                            axl[k].append(v)
    return {k: np.array(v) for k, v in axo.items()}, {k: np.array(v) for k, v in axl.items()}


@pytest.mark.skipif(not orc.ref_available(), reason="needs oracle/_ref (the reference compiled in place)")
def test_coupler_xgrid_equals_the_reference_loop(fg, gpu_ok):
    R = orc.ref()
    L = R                                                        # (argument types of the three integrals)
    L.poly_ctrlon.argtypes = [DP, DP, C.c_int, C.c_double]
    ni = 8
    lon, lat = fg.gnomonic_ed_corners(ni)
    atm = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    lo, la = fg.latlon_corners(30, 16)
    lnd = fg.GridConfig(30, 16, lo, la)
    to, ta = fg.tripolar_corners(40, 24)
    ocn = fg.GridConfig(40, 24, to, ta)
    jj, ii = np.meshgrid(np.arange(24), np.arange(40), indexing="ij")
    omask = np.where((ii // 5 + jj // 4) % 3 == 0, 0.0, 1.0)     # land blocks, sea, and a few coastal cells with fractions
    omask[(ii + 2 * jj) % 11 == 0] = 0.35
    omask[(2 * ii + jj) % 13 == 0] = 0.99995                     # land fraction below MIN_AREA_FRAC
    got = fg.coupler_xgrid(atm, lnd, ocn, omask, interp_order=2)
    axo, axl = _reference_coupler(R, atm, lnd, ocn, omask, 2)
    assert axo["area"].size > 1000 and axl["area"].size > 300
    for name, ref, mine, keys in (("axo", axo, got["axo"], ("ta", "ia", "ja", "io", "jo")), ("axl", axl, got["axl"], ("ta", "ia", "ja", "il", "jl"))):
        for k in keys:
            assert np.array_equal(np.asarray(ref[k], dtype=np.int64), np.asarray(mine[k], dtype=np.int64)), (name, k)
        assert np.max(np.abs(mine["area"] - ref["area"]) / ref["area"]) < 1e-10, name
        if orc.host_has_fma():
            for k in ("area", "clon", "clat"):
                assert np.array_equal(_bits(mine[k]), _bits(ref[k])), (name, k)
