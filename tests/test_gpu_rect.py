"""The rectilinear-target fast path of the legacy search (k_rect_tables / k_candidates_rect / k_clip_quad<., RECT>,
csrc/xgrid_kernels.hip) against the generic bins path: candidate rows / columns by index arithmetic on the two axes of the target
(get_output_grid_by_size grids, fregrid_util.c:588-654) must give the SAME plan bit for bit -- exchange-cell lists, areas,
centroid integrals, per-source-cell sums, destination cell records -- and a target that is not rectilinear must fall back to the
generic path inside the same call.  Oracle parity of the path itself: every other -m gpu test now runs through it (lat-lon
targets), e.g. tests/test_gpu_xgrid.py and tests/test_gpu_pipeline.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
D2R = np.pi / 180


def latlon_window(lon0, lon1, lat0, lat1, nlon, nlat, stretch=False):
    """corner arrays of a regular (or smoothly stretched) lat-lon window, built like get_output_grid_by_size builds them"""
    i = np.arange(nlon + 1, dtype=np.float64); j = np.arange(nlat + 1, dtype=np.float64)
    lon = (lon0 + i * ((lon1 - lon0) / nlon)) * D2R
    lat = (lat0 + j * ((lat1 - lat0) / nlat)) * D2R
    if stretch:                                   # still rectilinear, no longer uniform (the axis guess must fall back to its search)
        lon = lon[0] + (lon[-1] - lon[0]) * ((i / nlon) ** 1.7)
        lat = lat[0] + (lat[-1] - lat[0]) * (0.5 - 0.5 * np.cos(np.pi * j / nlat))
    return np.ascontiguousarray(np.broadcast_to(lon[None, :], (nlat + 1, nlon + 1))), np.ascontiguousarray(np.broadcast_to(lat[:, None], (nlat + 1, nlon + 1)))


def plan_dump(fg, order, grids, gout, masks, rect, cull=False):
    L = fg.lib()
    L.fg_set_search_rect(1 if rect else 0)
    L.fg_set_search_cull(1 if cull else 0)
    try:
        p = fg.XgridPlan.create(order, grids, gout, masks=masks)
        x = p.get_xgrid()                          # (before finalize: c1 / c2 are the centroid integrals)
        a_in, a_out = p.get_cell_area(gout.nx * gout.ny)
        out = dict(x, a_in=a_in, a_out=a_out, n=p.nxgrid, stats=p.stats())
        out["dst"] = p.get_cell_struct(1, gout.nx * gout.ny)
        if order == 2:
            import torch
            t = torch.empty(3 * p.ncells_in, dtype=torch.float64, device="cuda:0")
            p.copy_cell_sums(t); out["sums"] = t.cpu().numpy()
        p.destroy()
        return out
    finally:
        L.fg_set_search_rect(1); L.fg_set_search_cull(0)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def assert_same_plan(r, g, order, what):
    assert r["n"] == g["n"] > 0, what
    for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
        assert np.array_equal(r[k], g[k]), (what, k)
    keys = ["area", "a_out"] + (["c1", "c2", "sums"] if order == 2 else [])
    for k in keys:
        assert same_bits(r[k], g[k]), (what, k)
    for k in ("lat_min", "lat_max", "lon_min", "lon_max", "lon_avg", "nvert", "vlon", "vlat"):     # the tables spell out the generic records
        assert same_bits(r["dst"][k], g["dst"][k]), (what, "dst", k)


CASES = [
    # (name, source, target window (lon0, lon1, lat0, lat1, nlon, nlat), stretched)
    ("C48 -> global 2 deg", "c48", (0, 360, -90, 90, 180, 90), False),
    ("C48 -> 144x90", "c48", (0, 360, -90, 90, 144, 90), False),
    ("C48 -> global -180..180", "c48", (-180, 180, -90, 90, 120, 60), False),
    ("C48 -> regional window 230..310 x 15..65", "c48", (230, 310, 15, 65, 80, 50), False),
    ("C48 -> window across the date line -30..40", "c48", (-30, 40, -20, 30, 70, 50), False),
    ("C48 -> coarse 10 deg (big destination cells)", "c48", (0, 360, -90, 90, 36, 18), False),
    ("C24 -> fine 0.5 deg (every source cell heavy or big)", "c24", (0, 360, -90, 90, 720, 360), False),
    ("C48 -> stretched axes", "c48", (0, 360, -88, 88, 150, 80), True),
    ("tripolar -> 1 deg", "tri", (0, 360, -90, 90, 360, 180), False),
    ("lat-lon -> lat-lon (aligned edges)", "ll", (0, 360, -90, 90, 90, 45), False),
]


def source_grids(fg, kind):
    if kind in ("c48", "c24"):
        ni = int(kind[1:])
        lon, lat = fg.gnomonic_ed_corners(ni)
        return [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    if kind == "tri":
        lon, lat = fg.tripolar_corners(120, 90)
        return [fg.GridConfig(120, 90, lon, lat)]
    lo, la = fg.latlon_corners(180, 90)
    return [fg.GridConfig(180, 90, lo, la)]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("order", [1, 2])
def test_rectilinear_path_equals_the_generic_path(fg, gpu_ok, case, order):
    name, kind, win, stretch = case
    grids = source_grids(fg, kind)
    lo, la = latlon_window(*win, stretch=stretch)
    gout = fg.GridConfig(win[4], win[5], lo, la)
    r = plan_dump(fg, order, grids, gout, None, True)
    g = plan_dump(fg, order, grids, gout, None, False)
    assert g["stats"]["bins"] > 0 and r["stats"]["bins"] == 0, "the two searches must really take different paths"
    assert_same_plan(r, g, order, name)


def test_rectilinear_path_with_masks_and_culling(fg, gpu_ok):
    grids = source_grids(fg, "c48")
    rng = np.random.default_rng(5)
    masks = [(rng.random((48, 48)) > 0.3).astype(np.float64) for _ in range(6)]
    masks[2] = None
    lo, la = latlon_window(0, 360, -90, 90, 144, 90)
    for j0, j1 in ((0, 90), (0, 30), (30, 61), (61, 90)):                      # whole target and three bands of it, culling on
        gout = fg.GridConfig(144, j1 - j0, np.ascontiguousarray(lo[j0:j1 + 1]), np.ascontiguousarray(la[j0:j1 + 1]))
        r = plan_dump(fg, 2, grids, gout, masks, True, cull=True)
        g = plan_dump(fg, 2, grids, gout, masks, False, cull=False)
        assert r["stats"]["bins"] == 0
        for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
            assert np.array_equal(r[k], g[k]), k
        for k in ("area", "c1", "c2", "sums", "a_out"):
            assert same_bits(r[k], g[k]), k


def test_targets_that_are_not_rectilinear_fall_back(fg, gpu_ok):
    """A cubed-sphere tile, a rotated lat-lon grid and a lat-lon grid with ONE corner moved by one ulp as targets: the device
    check refuses them and the same call returns the generic path's plan."""
    src = source_grids(fg, "ll")
    lon, lat = fg.gnomonic_ed_corners(24)
    lo, la = latlon_window(0, 360, -90, 90, 72, 36)
    lo2 = lo.copy(); lo2[17, 23] = np.nextafter(lo2[17, 23], 10.0)
    la3 = la.copy(); la3[5, 60] = np.nextafter(la3[5, 60], 10.0)
    for name, gout in (("cubed-sphere tile", fg.GridConfig(24, 24, lon[1], lat[1])),
                       ("one longitude off by an ulp", fg.GridConfig(72, 36, lo2, la)),
                       ("one latitude off by an ulp", fg.GridConfig(72, 36, lo, la3))):
        r = plan_dump(fg, 2, src, gout, None, True)
        assert r["stats"]["bins"] > 0, name
        g = plan_dump(fg, 2, src, gout, None, False)
        assert_same_plan(r, g, 2, name)
