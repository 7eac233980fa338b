"""GPU parity of the great-circle path (create_xgrid_great_circle semantics) against the oracle
(gc_oracle.c, itself pinned bit for bit to the compiled reference): exchange-cell lists identical, areas within 1e-10
relative -- and in fact bit-identical except where the x87 fpatan inside glibc's acosl rounds differently from the
device's correctly rounded replacement (csrc/fp80.h), which the tests quantify."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _cases(fg):
    c24 = fg.gnomonic_ed_corners(24)
    c12 = fg.gnomonic_ed_corners(12)
    return {
        "c24_tile1_144x90": (24, 24, 144, 90, c24[0][0], c24[1][0]) + fg.latlon_corners(144, 90),
        "c24_tile3_polar_144x90": (24, 24, 144, 90, c24[0][2], c24[1][2]) + fg.latlon_corners(144, 90),
        "latlon_aligned_2x": (36, 18, 72, 36) + fg.latlon_corners(36, 18) + fg.latlon_corners(72, 36),
        "latlon_regional_offset": (30, 20, 45, 33) + fg.latlon_corners(30, 20, 10., 70., -30., 30.) + fg.latlon_corners(45, 33, 0., 90., -40., 40.),
        "latlon_to_cubed_polar": (40, 20, 12, 12) + fg.latlon_corners(40, 20) + (c12[0][5], c12[1][5]),
        "tripolar_to_cubed": (45, 27, 12, 12) + fg.tripolar_corners(45, 27) + (c12[0][2], c12[1][2]),
    }


@pytest.mark.parametrize("name", ["c24_tile1_144x90", "c24_tile3_polar_144x90", "latlon_aligned_2x", "latlon_regional_offset",
                                  "latlon_to_cubed_polar", "tripolar_to_cubed"])
def test_create_xgrid_great_circle_vs_oracle(fg, gpu_ok, name):
    args = _cases(fg)[name]
    o = orc.orc_create_xgrid_gc(*args)
    n, i_in, j_in, i_out, j_out, area, clon, clat = fg.create_xgrid_great_circle(*args)
    assert n == o["n"] and n > 0
    assert np.array_equal(i_in, o["i_in"]) and np.array_equal(j_in, o["j_in"])
    assert np.array_equal(i_out, o["i_out"]) and np.array_equal(j_out, o["j_out"])
    rel = np.abs(area - o["area"]) / o["area"]
    assert rel.max() < RTOL
    same = np.mean(_bits(area) == _bits(o["area"]))
    assert same > 0.98, same                       # the rest: one final double rounding of an angle (fpatan vs exact)
    assert not clon.any() and not clat.any()       # create_xgrid.c:1446-1447
    ca = fg.get_grid_great_circle_area(args[0], args[1], args[4], args[5])
    cref = orc.orc_get_grid_gc_area(args[0], args[1], args[4], args[5])
    assert np.max(np.abs(ca - cref) / np.abs(cref)) < RTOL
    assert np.mean(_bits(ca) == _bits(cref)) > 0.98


def test_clip_vertices_bitwise(fg, gpu_ok):
    """clip_2dx2d_great_circle vertex lists: the soft-x87 intersection solve reproduces the reference's vertices bit for
    bit (no acos involved in the vertices)."""
    import ctypes as C
    O = orc.oracle()
    lon, lat = fg.gnomonic_ed_corners(24)
    lo, la = fg.latlon_corners(144, 90)
    x1, y1, z1 = fg.latlon2xyz(lon[2], lat[2])
    x2, y2, z2 = fg.latlon2xyz(lo, la)
    ox, oy, oz = (np.empty(25 * 25) for _ in range(3))
    O.orc_latlon2xyz(25 * 25, orc._dp(orc.f64(lon[2]).ravel()), orc._dp(orc.f64(lat[2]).ravel()), orc._dp(ox), orc._dp(oy), orc._dp(oz))
    assert np.array_equal(_bits(x1), _bits(ox)) and np.array_equal(_bits(y1), _bits(oy)) and np.array_equal(_bits(z1), _bits(oz))
    cell = lambda x, nxp, i, j: np.array([x[j * nxp + i], x[(j + 1) * nxp + i], x[(j + 1) * nxp + i + 1], x[j * nxp + i + 1]])
    gc = orc.orc_create_xgrid_gc(24, 24, 144, 90, lon[2], lat[2], lo, la)
    idx = np.arange(0, gc["n"], 3)
    a = np.stack([np.stack([cell(v, 25, int(gc["i_in"][k]), int(gc["j_in"][k])) for v in (x1, y1, z1)], axis=1) for k in idx])
    b = np.stack([np.stack([cell(v, 145, int(gc["i_out"][k]), int(gc["j_out"][k])) for v in (x2, y2, z2)], axis=1) for k in idx])
    n_out, verts, area = fg.gc_clip_batch(a, b)
    nbad_area = 0
    for q, k in enumerate(idx):
        aa = [np.ascontiguousarray(a[q][:, ax]) for ax in range(3)]
        bb = [np.ascontiguousarray(b[q][:, ax]) for ax in range(3)]
        oo = [np.zeros(50) for _ in range(3)]
        no = O.orc_clip_2dx2d_great_circle(*[orc._dp(v) for v in aa], 4, *[orc._dp(v) for v in bb], 4, *[orc._dp(v) for v in oo])
        assert no == n_out[q] and no >= 3
        for ax in range(3):
            assert np.array_equal(_bits(verts[q, :no, ax]), _bits(oo[ax][:no]))
        ar = O.orc_great_circle_area(no, *[orc._dp(v) for v in oo])
        assert abs(ar - area[q]) <= RTOL * abs(ar)
        nbad_area += ar != area[q]
    assert nbad_area < 0.02 * len(idx)
    # an antipodal cell (whose edge planes do cross the first cell's chords) is rejected by the reference's bounding box
    # (create_xgrid.c:1508-1528) and so it is here
    anti = -a[0][::-1]
    n_far, _, _ = fg.gc_clip_batch(a[:1], anti[None])
    assert n_far[0] == 0


def test_setup_conserve_interp_great_circle_and_sweep(fg, gpu_ok, capsys):
    """opcode GREAT_CIRCLE through the mirror API: whole-tile search for all six tiles, first-order sweep, conservation."""
    ni, nlon, nlat = 16, 48, 24
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grid_in = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    grid_out = [fg.GridConfig(nlon, nlat, lo, la)]
    interp = [fg.InterpConfig()]
    fg.setup_conserve_interp(6, grid_in, 1, grid_out, interp, fg.CONSERVE_ORDER1 | fg.GREAT_CIRCLE)
    tot = 0
    for t in range(6):
        o = orc.orc_create_xgrid_gc(ni, ni, nlon, nlat, lon[t], lat[t], lo, la)
        sel = interp[0].t_in == t
        assert sel.sum() == o["n"]
        assert np.array_equal(interp[0].i_in[sel], o["i_in"]) and np.array_equal(interp[0].j_out[sel], o["j_out"])
        assert np.max(np.abs(interp[0].area[sel] - o["area"]) / o["area"]) < RTOL
        tot += o["n"]
    assert interp[0].nxgrid == tot
    R = 6371000.0
    assert abs(interp[0].area.sum() / (4 * np.pi * R * R) - 1) < 1e-9          # great-circle cells tile the sphere
    # cell areas handed back are the great-circle ones (get_input_output_cell_area with GREAT_CIRCLE)
    assert np.max(np.abs(grid_in[2].cell_area - orc.orc_get_grid_gc_area(ni, ni, lon[2], lat[2])) / grid_in[2].cell_area) < RTOL
    rng = np.random.default_rng(2)
    field_in = [fg.FieldConfig(data=rng.standard_normal((1, ni, ni)) + 3.0, var=[fg.VarConfig(name="t", interp_method=fg.CONSERVE_ORDER1)])
                for _ in range(6)]
    field_out = [fg.FieldConfig()]
    gin, gout = fg.do_scalar_conserve_interp(interp, 0, 6, grid_in, 1, grid_out, field_in, field_out,
                                             fg.CONSERVE_ORDER1 | fg.CHECK_CONSERVE, 1)
    assert abs(gout - gin) < 1e-10 * abs(gin)
    with pytest.raises(ValueError):
        fg.setup_conserve_interp(6, grid_in, 1, grid_out, [fg.InterpConfig()], fg.CONSERVE_ORDER2 | fg.GREAT_CIRCLE)


def test_conserve_interp_great_circle_b1(fg, gpu_ok):
    """interp.c:312 drop-in against the same formula on the oracle's exchange cells (sum of area fractions per
    destination cell, in exchange-cell order)."""
    ni, nlon, nlat = 12, 36, 18
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    rng = np.random.default_rng(9)
    data = rng.standard_normal(ni * ni) + 2.0
    mask = np.ones(ni * ni); mask[5] = 0.0
    got = fg.conserve_interp_great_circle(ni, ni, nlon, nlat, lon[0], lat[0], lo, la, mask, data)
    o = orc.orc_create_xgrid_gc(ni, ni, nlon, nlat, lon[0], lat[0], lo, la, mask=mask)
    d = o["j_out"].astype(np.int64) * nlon + o["i_out"]
    dst_area = np.zeros(nlon * nlat)
    for k in range(o["n"]):
        dst_area[d[k]] += o["area"][k]
    ref = np.zeros(nlon * nlat)
    for k in range(o["n"]):
        ref[d[k]] += data[o["j_in"][k] * ni + o["i_in"][k]] * (o["area"][k] / dst_area[d[k]])
    assert np.max(np.abs(got - ref)) < 1e-12 * np.max(np.abs(ref))
    assert np.count_nonzero(got) == np.count_nonzero(ref) > 0


def test_great_circle_degenerate_cases(fg, gpu_ok):
    """Edge cases against the oracle: identical grids (every edge pair coincident: u snapped to 0/1 everywhere), disjoint
    regional grids (no exchange cells), single cells, and a masked-out source."""
    lo, la = fg.latlon_corners(24, 12)
    o = orc.orc_create_xgrid_gc(24, 12, 24, 12, lo, la, lo, la)
    r = fg.create_xgrid_great_circle(24, 12, 24, 12, lo, la, lo, la)
    assert r[0] == o["n"] and np.array_equal(r[1], o["i_in"]) and np.array_equal(r[3], o["i_out"]) and np.array_equal(r[4], o["j_out"])
    assert np.max(np.abs(r[5] - o["area"]) / o["area"]) < RTOL
    assert r[0] >= 24 * 12                              # every cell at least meets itself
    lon, lat = fg.gnomonic_ed_corners(8)
    o = orc.orc_create_xgrid_gc(8, 8, 8, 8, lon[3], lat[3], lon[3], lat[3])
    r = fg.create_xgrid_great_circle(8, 8, 8, 8, lon[3], lat[3], lon[3], lat[3])
    assert r[0] == o["n"] and np.array_equal(r[3], o["i_out"]) and np.array_equal(r[4], o["j_out"])
    assert np.max(np.abs(r[5] - o["area"]) / o["area"]) < RTOL
    # disjoint regions
    lo1, la1 = fg.latlon_corners(6, 6, 10.0, 40.0, 10.0, 40.0)
    lo2, la2 = fg.latlon_corners(5, 5, 100.0, 140.0, -40.0, -10.0)
    assert orc.orc_create_xgrid_gc(6, 6, 5, 5, lo1, la1, lo2, la2)["n"] == 0
    assert fg.create_xgrid_great_circle(6, 6, 5, 5, lo1, la1, lo2, la2)[0] == 0
    # single cells, partial overlap
    lo3, la3 = fg.latlon_corners(1, 1, 10.0, 20.0, 10.0, 20.0)
    lo4, la4 = fg.latlon_corners(1, 1, 15.0, 30.0, 5.0, 15.0)
    o = orc.orc_create_xgrid_gc(1, 1, 1, 1, lo3, la3, lo4, la4)
    r = fg.create_xgrid_great_circle(1, 1, 1, 1, lo3, la3, lo4, la4)
    assert r[0] == o["n"] == 1 and abs(r[5][0] - o["area"][0]) < RTOL * o["area"][0]
    # fully masked source
    r = fg.create_xgrid_great_circle(6, 6, 6, 6, lo1, la1, lo1, la1, mask_in=np.zeros(36))
    assert r[0] == 0


@pytest.mark.parametrize("ni,nlon,nlat", [(24, 144, 90), (96, 360, 180), (48, 1440, 720)])
def test_three_pass_clip_equals_the_one_kernel_clip(fg, gpu_ok, ni, nlon, nlat):
    """The great-circle clip runs as three passes (k_gc_screen / k_gc_solve / k_gc_walk) for ordinary pairs and as the
    one-kernel clip (the version pinned to the oracle above) for the rest.  Here every pair goes through each of the two
    and the plans must be bit-identical; only a small fraction of the pairs may be handed back by the three passes."""
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    res = []
    try:
        # one kernel | three passes | three passes in 3 chunks of source cells | three passes with a task buffer 64 times too small
        for split, chunks in ((0, 1), (1, 1), (1, 3), (2, 1)):
            fg.lib().fg_set_gc_split(split)
            fg.lib().fg_set_search_chunks(chunks)
            plan = fg.XgridPlan.create_great_circle(grids, fg.GridConfig(nlon, nlat, lo, la))
            plan.finalize()
            res.append((plan.get_xgrid(), plan.stats()))
            plan.destroy()
    finally:
        fg.lib().fg_set_gc_split(1)
        fg.lib().fg_set_search_chunks(0)
    a, sa = res[0]
    assert len(a["area"]) > 0 and sa["deferred"] == 0
    for b, sb in res[1:]:
        assert len(a["area"]) == len(b["area"])
        for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
            assert np.array_equal(a[k], b[k])
        assert np.array_equal(_bits(a["area"]), _bits(b["area"]))
        assert sa["below"] == sb["below"] and sa["borderline"] == sb["borderline"]
    assert res[1][1]["deferred"] < 0.1 * sa["pairs"] + 4000   # pole cells, tile-edge cells on lat-lon lines, snapped intersections
    assert res[3][1]["deferred"] > 0.1 * sa["pairs"]          # the overflow path really ran


def _rotated(lon, lat, ax, ang):
    """corner arrays rotated about the unit axis `ax` by `ang` (a curvilinear grid whose edges cross everything at odd angles)"""
    x, y, z = np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)
    v = np.stack([x, y, z], -1)
    ax = np.asarray(ax, float) / np.linalg.norm(ax)
    c, s = np.cos(ang), np.sin(ang)
    r = v * c + np.cross(ax, v) * s + ax * (v @ ax)[..., None] * (1 - c)
    lo = np.arctan2(r[..., 1], r[..., 0])
    lo[lo < 0] += 2 * np.pi
    return np.ascontiguousarray(lo), np.ascontiguousarray(np.arcsin(np.clip(r[..., 2], -1, 1)))


def test_three_pass_clip_on_awkward_grid_pairs(fg, gpu_ok):
    """Pairs of grids chosen to sit on the branches of the three-pass clip: edges and corners that almost coincide (shifts of
    1e-9 .. 1e-5 rad: snapped, nearly snapped and borderline-inside cases), generic crossings at odd angles (rotated grids),
    coarse against fine, tripolar folds, regional windows.  Three passes == one-kernel clip bit for bit; the small cases also
    against the oracle."""
    lo72, la72 = fg.latlon_corners(72, 36)
    reg = fg.latlon_corners(40, 30, 20.0, 100.0, -35.0, 40.0)
    c16 = fg.gnomonic_ed_corners(16)
    tri = fg.tripolar_corners(60, 40)
    cases = []
    for sh in (1e-9, 3e-8, 1e-6, 1e-5):
        cases.append((72, 36, 72, 36, lo72, la72, np.ascontiguousarray(lo72 + sh), la72))
    cases.append((40, 30, 40, 30) + reg + _rotated(*reg, (0.3, -0.5, 0.8), 0.37))
    cases.append((72, 36, 40, 30, lo72, la72) + _rotated(*reg, (1.0, 0.2, 0.1), 1.1))
    cases.append((16, 16, 40, 30, c16[0][1], c16[1][1]) + reg)
    cases.append((16, 16, 16, 16, c16[0][0], c16[1][0]) + _rotated(c16[0][0], c16[1][0], (0.1, 0.9, 0.4), 0.05))
    cases.append((60, 40, 72, 36) + tri + (lo72, la72))
    def run(args):
        """plan API (errors come back as exceptions instead of the reference's exit): (xgrid dict | None, message)"""
        nxi, nyi, nxo, nyo, loi, lai, loo, lao = args
        try:
            plan = fg.XgridPlan.create_great_circle([fg.GridConfig(nxi, nyi, loi, lai)], fg.GridConfig(nxo, nyo, loo, lao))
        except Exception as e:                       # the reference's own fatal checks (mpp_error) on this input
            return None, str(e)
        x = plan.get_xgrid() if plan.nxgrid else {"area": np.zeros(0)}
        plan.destroy()
        return x, ""

    nok = 0
    for ci, args in enumerate(cases):
        outs = []
        try:
            for split in (0, 1):
                fg.lib().fg_set_gc_split(split)
                outs.append(run(args))
        finally:
            fg.lib().fg_set_gc_split(1)
        (a, ea), (b, eb) = outs
        assert ea == eb, (ci, ea, eb)                # same fatal check, same message -- or none
        if a is None:
            continue
        nok += 1
        assert len(a["area"]) == len(b["area"]), ci
        if len(a["area"]) == 0:
            continue
        for k in ("i_in", "j_in", "i_out", "j_out"):
            assert np.array_equal(a[k], b[k]), (ci, k)
        assert np.array_equal(_bits(a["area"]), _bits(b["area"])), ci
        if args[0] * args[1] * args[2] * args[3] <= 40 * 30 * 40 * 30:
            o = orc.orc_create_xgrid_gc(*args)
            assert len(b["area"]) == o["n"], ci
            assert np.array_equal(b["i_in"], o["i_in"]) and np.array_equal(b["j_in"], o["j_in"]), ci
            assert np.array_equal(b["i_out"], o["i_out"]) and np.array_equal(b["j_out"], o["j_out"]), ci
            assert np.max(np.abs(b["area"] - o["area"]) / o["area"]) < RTOL, ci
    assert nok >= 5
