"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares,
the oracle reproduces the committed golden vectors (made by the compiled reference), host logic."""
import json
import os
import re

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
D2R = np.pi / 180


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_library_exports_every_declared_symbol(fg):
    import ctypes
    L = fg.lib()
    hdr = open(os.path.join(ROOT, "include", "fregrid_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", hdr)) - {"defined", "sizeof"}
    declared = {d for d in declared if not d.isupper()}
    assert len(declared) >= 40
    missing = [d for d in sorted(declared) if not hasattr(L, d)]
    assert not missing, missing
    assert set(fg._lib.EXPORTS) <= declared | {"fg_pool_release"}
    assert L.get_maxxgrid() == 5000000          # create_xgrid.h:22-28, serial build


def test_no_gpu_fails_loudly(fg):
    """The product path must not fall back to the CPU: without a device every compute entry raises."""
    if fg.lib().fg_device_count() > 0:
        pytest.skip("a GPU is present")
    lon, lat = fg.gnomonic_ed_corners(4)
    lo, la = fg.latlon_corners(8, 4)
    with pytest.raises(fg.FregridHipError):
        fg.create_xgrid_2dx2d_order1(4, 4, 8, 4, lon[0], lat[0], lo, la)
    with pytest.raises(fg.FregridHipError):
        fg.XgridPlan.create(1, [fg.GridConfig(4, 4, lon[0], lat[0])], fg.GridConfig(8, 4, lo, la))
    with pytest.raises(fg.FregridHipError):
        fg.create_xgrid_great_circle(4, 4, 8, 4, lon[0], lat[0], lo, la)
    with pytest.raises(fg.FregridHipError):
        fg.XgridPlan.create_great_circle([fg.GridConfig(4, 4, lon[0], lat[0])], fg.GridConfig(8, 4, lo, la))
    with pytest.raises(fg.FregridHipError):
        fg.create_xgrid_box(True, 1, np.linspace(0, 1, 5), np.linspace(0, 1, 3), 4, 4, lon[0], lat[0])
    with pytest.raises(fg.FregridHipError):
        fg.gc_clip_batch(np.zeros((1, 4, 3)), np.zeros((1, 4, 3)))
    # the plan-level C entry points report the missing device through their return code (no exit, no CPU path)
    import ctypes as C
    h = C.c_void_p()
    nx = (C.c_int * 1)(4)
    dpt = C.POINTER(C.c_double)
    a = np.ascontiguousarray(lon[0]).ravel(); b = np.ascontiguousarray(lat[0]).ravel()
    lo1 = np.ascontiguousarray(lo).ravel(); la1 = np.ascontiguousarray(la).ravel()
    lonp = (dpt * 1)(a.ctypes.data_as(dpt)); latp = (dpt * 1)(b.ctypes.data_as(dpt))
    rc = fg.lib().fg_plan_create_great_circle(1, nx, nx, lonp, latp, None, 8, 4, lo1.ctypes.data_as(dpt), la1.ctypes.data_as(dpt), 0, C.byref(h))
    assert rc == -2 and b"device" in fg.lib().fg_last_error()          # FG_ERR_HIP


def test_grid_generator_golden(fg):
    g = np.load(os.path.join(GOLD, "c48_grid.npz"))
    lon, lat = fg.gnomonic_ed_corners(48)
    assert np.array_equal(_bits(lon), _bits(g["lon"])) and np.array_equal(_bits(lat), _bits(g["lat"]))
    lo, la = fg.latlon_corners(180, 90)
    assert lo.shape == (91, 181) and abs(lo[0, -1] - 2 * np.pi) < 1e-15 and abs(la[-1, 0] - np.pi / 2) < 1e-15
    assert np.array_equal(lo[0], lo[-1]) and np.array_equal(la[:, 0], la[:, -1])
    # get_output_grid_by_size with center_y == 0 (fregrid_util.c:606-610)
    lo2, la2 = fg.latlon_corners(4, 3, 0, 360, -90, 90, center_y=False)
    assert abs(la2[0, 0] - (-90 - 45) * D2R) < 1e-15


@pytest.mark.parametrize("key,order,nlon,nlat,tile", [("o1_180x90_t1", 1, 180, 90, 0), ("o1_180x90_t3", 1, 180, 90, 2),
                                                      ("o2_144x90_t1", 2, 144, 90, 0), ("o2_144x90_t3", 2, 144, 90, 2)])
def test_oracle_reproduces_reference_golden(fg, key, order, nlon, nlat, tile):
    g = np.load(os.path.join(GOLD, "c48_xgrid.npz"))
    grid = np.load(os.path.join(GOLD, "c48_grid.npz"))
    lo, la = fg.latlon_corners(nlon, nlat)
    r = orc.orc_create_xgrid(order, 48, 48, nlon, nlat, grid["lon"][tile], grid["lat"][tile], lo, la)
    assert r["n"] == len(g[key + "_area"])
    for k in ("i_in", "j_in", "i_out", "j_out"):
        assert np.array_equal(r[k], g[f"{key}_{k}"]), k
    for k in ("area",) + (("clon", "clat") if order == 2 else ()):
        assert np.array_equal(_bits(r[k]), _bits(g[f"{key}_{k}"])), k


def test_oracle_clip_known_answers():
    """The legacy-clip cases of the reference's embedded test main (create_xgrid.c:2825-3015) and the
    statements printed with them (:3125-3130): 'Must result n_out=5', 'the second box', ..."""
    L = orc.oracle()
    dp = orc.dp
    P = lambda v: v.ctypes.data_as(dp)
    cases = json.load(open(os.path.join(GOLD, "clip_cases.json")))["cases"]
    areas = {}
    for c in cases:
        pad = lambda v: np.array(list(v) + [0.0] * (60 - len(v)))
        x1, y1 = pad(np.array(c["lon1_deg"]) * D2R), pad(np.array(c["lat1_deg"]) * D2R)
        x2, y2 = pad(np.array(c["lon2_deg"]) * D2R), pad(np.array(c["lat2_deg"]) * D2R)
        xo, yo = np.zeros(60), np.zeros(60)
        n = L.orc_clip_2dx2d(P(x1), P(y1), len(c["lon1_deg"]), P(x2), P(y2), len(c["lon2_deg"]), P(xo), P(yo))
        ref = c["ref"]
        assert n == ref["n_clip"], c["case"]
        assert np.array_equal(_bits(xo[:n]), _bits(np.array(ref["clip_lon"]))) and np.array_equal(_bits(yo[:n]), _bits(np.array(ref["clip_lat"])))
        n2 = L.orc_fix_lon(P(x2), P(y2), len(c["lon2_deg"]), np.pi)
        nf = L.orc_fix_lon(P(xo), P(yo), n, np.pi)
        assert nf == ref["n_out_fixed"] and n2 == ref["n2_fixed"]
        a_out = L.orc_poly_area(P(xo), P(yo), nf)
        a2 = L.orc_poly_area(P(x2), P(y2), n2)
        assert a_out == ref["area_out"] and a2 == ref["area2"]
        areas[c["case"]] = a_out
        e = c["expect"]
        if e == "n_out=5":
            assert n == 5
        elif e == "n_out=4":
            assert n == 4
        elif e == "n_out=0":
            assert n == 0
        elif e.startswith("second box") or e.startswith("same box"):
            assert abs(a_out - a2) <= 1e-9 * a2
    assert abs(areas[22] - areas[23]) <= 1e-9 * areas[22]
    assert abs(areas[24] - areas[25]) <= 1e-9 * areas[24] and abs(areas[24] - areas[26]) <= 1e-9 * areas[24]


def test_oracle_counts_c48(fg):
    counts = json.load(open(os.path.join(GOLD, "counts.json")))
    lon, lat = fg.gnomonic_ed_corners(48)
    lo, la = fg.latlon_corners(180, 90)
    tot = 0
    for t in range(6):
        n = orc.orc_create_xgrid(1, 48, 48, 180, 90, lon[t], lat[t], lo, la)["n"]
        exp = counts["C48->180x90 o1"]["tiles_36" if t in (2, 5) else "tiles_1245"]
        assert n == exp
        tot += n
    assert tot == counts["C48->180x90 o1"]["total"]


def test_oracle_setup_and_apply_conserve(fg):
    """Oracle-level end to end on a small case (C16 -> 36x18, order 2): centroid distances sum to zero per source
    cell, the sweep conserves the flux to the reference's own closure (~1e-9), constant fields stay constant."""
    ni = 16
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(36, 18)
    x = orc.orc_setup(2, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(36, 18, lo, la)])
    assert x["n"] > 0 and np.all(np.diff(x["t_in"]) >= 0)
    key = (x["t_in"].astype(np.int64) * ni * ni + x["j_in"] * ni + x["i_in"]) * 36 * 18 + x["j_out"] * 36 + x["i_out"]
    assert np.all(np.diff(key) > 0)            # canonical order: tile, j_in, i_in, then destination index
    s = x["t_in"].astype(np.int64) * ni * ni + x["j_in"] * ni + x["i_in"]
    for arr in (x["di"], x["dj"]):
        tot = np.bincount(s, weights=arr * x["area"], minlength=6 * ni * ni)
        assert np.max(np.abs(tot)) < 1e-3      # m^2 rad, vs cell areas ~1e11 m^2
    const = [np.full((ni + 2) * (ni + 2), 3.25) for _ in range(6)]
    g = [np.zeros(ni * ni) for _ in range(6)]
    out, gs = orc.orc_apply(2, x, [ni] * 6, [ni] * 6, const, g, g, None, False, 0.0, 36, 18, 1)
    assert np.max(np.abs(out - 3.25)) < 1e-14
    earth = 4 * np.pi * 6371000.0 ** 2
    assert abs(gs / 3.25 - earth) / earth < 5e-9


def test_tripolar_generator_properties(fg):
    """fg_tripolar_corners (unpinned restatement of create_tripolar_grid): regular below the join latitude, cap cells
    close on the spherical cap, the top row is the fold (mirror-symmetric about lon_start+90 / +270)."""
    nlon, nlat = 72, 43
    lon, lat = fg.tripolar_corners(nlon, nlat)
    lonr, latr = fg.latlon_corners(nlon, nlat, -280.0, 80.0, -82.0, 90.0)
    yb = np.degrees(latr[:, 0])
    jj = int(np.argmin(np.abs(yb - 65.0)))
    assert np.array_equal(lon[:jj], lonr[:jj]) or np.max(np.abs(lon[:jj] - lonr[:jj])) < 1e-13
    assert np.max(np.abs(lat[:jj + 1] - latr[:jj + 1])) < 1e-13
    assert np.all(lat[jj:] >= lat[jj, 0] - 1e-12) and np.all(lat <= np.pi / 2)
    area = orc.orc_get_grid_area(nlon, nlat, lon, lat)
    assert area.min() > 0
    R = 6371000.0
    cap = 2 * np.pi * R * R * (1 + np.sin(np.radians(82.0)))
    assert abs(area.sum() / cap - 1) < 1e-9
    top_lat = lat[-1]
    assert np.max(np.abs(top_lat - top_lat[::-1])) < 1e-12          # fold symmetry
    half = nlon // 2
    assert np.max(np.abs(top_lat[:half + 1] - top_lat[:half + 1][::-1])) < 1e-12


def _small_case(fg, order):
    ni, nlon, nlat = 12, 36, 18
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    o = orc.orc_setup(order, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    rng = np.random.default_rng(5)
    h = 1 if order == 2 else 0
    data = [rng.standard_normal((1, ni + 2 * h, ni + 2 * h)) + 5.0 for _ in range(6)]
    gx = [rng.standard_normal((1, ni, ni)) for _ in range(6)]
    gy = [rng.standard_normal((1, ni, ni)) for _ in range(6)]
    return ni, nlon, nlat, lo, la, o, data, gx, gy


@pytest.mark.parametrize("order", [1, 2])
def test_oracle_ex_reduces_to_plain_branch(fg, order):
    """orc_do_scalar_conserve_interp_ex with every option off reproduces the plain-branch oracle bit for bit."""
    ni, nlon, nlat, lo, la, o, data, gx, gy = _small_case(fg, order)
    gm = [np.zeros((ni, ni), dtype=np.int32) for _ in range(6)]
    a, ga = orc.orc_apply(order, o, [ni] * 6, [ni] * 6, data, gx, gy, gm, False, 0.0, nlon, nlat, 1)
    rc, b, gb = orc.orc_apply_ex(order, o, [ni] * 6, [ni] * 6, data, gx, gy, gm, False, 0.0, nlon, nlat, 1)
    assert rc == 0 and np.array_equal(a.view(np.uint64), b.view(np.uint64)) and ga == gb


def test_oracle_monotone_limiter_bounds(fg):
    """Monotone branch: every remapped value stays within the extremes of the source field, and a field with zero
    gradients is untouched by the limiter (equals the first-order remap of the same weights)."""
    ni, nlon, nlat, lo, la, o, data, gx, gy = _small_case(fg, 2)
    big = [g * 3.0 for g in gx], [g * 3.0 for g in gy]
    rc, out, _ = orc.orc_apply_ex(2, o, [ni] * 6, [ni] * 6, data, big[0], big[1], None, False, 0.0, nlon, nlat, 1,
                                  monotonic=True)
    assert rc == 0
    lo_v = min(d.min() for d in data)
    hi_v = max(d.max() for d in data)
    assert out.min() >= lo_v - 1e-9 and out.max() <= hi_v + 1e-9
    rc, unl, _ = orc.orc_apply_ex(2, o, [ni] * 6, [ni] * 6, data, big[0], big[1], None, False, 0.0, nlon, nlat, 1)
    assert unl.max() > hi_v or unl.min() < lo_v or not np.array_equal(unl, out)
    zero = [np.zeros_like(g) for g in gx]
    rc, z, _ = orc.orc_apply_ex(2, o, [ni] * 6, [ni] * 6, data, zero, zero, None, False, 0.0, nlon, nlat, 1,
                                monotonic=True)
    rc2, p, _ = orc.orc_apply_ex(2, o, [ni] * 6, [ni] * 6, data, zero, zero, None, False, 0.0, nlon, nlat, 1)
    assert rc == 0 and rc2 == 0 and np.array_equal(z, p)


def test_oracle_sum_and_target_grid_branches(fg):
    """cell_methods=sum conserves the plain sum of the field; --target_grid rescales by covered/own cell area."""
    ni, nlon, nlat, lo, la, o, data, gx, gy = _small_case(fg, 1)
    ca = o["cell_area_in"]                                        # per-tile list (get_grid_area values)
    rc, out, gs = orc.orc_apply_ex(1, o, [ni] * 6, [ni] * 6, data, None, None, None, False, 0.0, nlon, nlat, 1,
                                   cell_methods_sum=True, cell_area_in=ca)
    assert rc == 0
    tot_in = sum(d.sum() for d in data)
    assert abs(out.sum() - tot_in) < 1e-8 * abs(tot_in)          # closure of the exchange grid itself ~1e-9
    cao = orc.orc_get_grid_area(nlon, nlat, lo, la)
    rc, t, _ = orc.orc_apply_ex(1, o, [ni] * 6, [ni] * 6, data, None, None, None, False, 0.0, nlon, nlat, 1,
                                target_grid=True, cell_area_out=cao)
    rc2, p, _ = orc.orc_apply_ex(1, o, [ni] * 6, [ni] * 6, data, None, None, None, False, 0.0, nlon, nlat, 1)
    assert rc == 0 and rc2 == 0
    assert np.max(np.abs(t / p - 1)) < 5e-3 and not np.array_equal(t, p)


@pytest.mark.parametrize("t", [0, 2])
def test_great_circle_oracle_reproduces_reference_golden(t):
    """tests/golden/gc_c24_xgrid.npz was written by the compiled reference (make_golden_gc.py); this check does not
    need oracle/_ref, so it also runs where the reference is absent."""
    g = np.load(os.path.join(GOLD, "gc_c24_xgrid.npz"))
    o = orc.orc_create_xgrid_gc(24, 24, 144, 90, g[f"lon_t{t}"], g[f"lat_t{t}"], g["lon_out"], g["lat_out"])
    assert o["n"] == len(g[f"area_t{t}"])
    for k in ("i_in", "j_in", "i_out", "j_out"):
        assert np.array_equal(o[k], g[f"{k}_t{t}"]), k
    assert np.array_equal(o["area"].view(np.uint64), g[f"area_t{t}"].view(np.uint64))
    ca = orc.orc_get_grid_gc_area(24, 24, g[f"lon_t{t}"], g[f"lat_t{t}"])
    assert np.array_equal(ca.view(np.uint64), g[f"cell_area_t{t}"].view(np.uint64))
