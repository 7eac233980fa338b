"""GPU parity for the whole path: setup_conserve_interp (search + centroid pass) and
do_scalar_conserve_interp (the sweep) through the C ABI, against the CPU oracle; plus edge cases
(regional target, curvilinear target incl. pole caps, masked/missing data, empty overlap, 1x1 grids)
and the polygon known-answer cases of the reference's embedded test main."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
D2R = np.pi / 180
RTOL = 1e-10


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def check_xgrid(x, o, order, finalized):
    assert len(x["area"]) == o["n"]
    for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
        assert np.array_equal(x[k], o[k]), k
    assert np.max(np.abs(x["area"] - o["area"]) / o["area"]) < RTOL
    # beyond the 1e-10 bar: the device evaluates sin/cos with the host libm's operation sequence (csrc/sincos_glibc.h), every
    # other operation is IEEE add/mul/div in the reference's order, so the areas carry the reference's bits
    if orc.host_has_fma():
        assert np.array_equal(_bits(x["area"]), _bits(o["area"]))
        if order == 2 and finalized:
            assert np.array_equal(_bits(x["c1"]), _bits(o["di"])) and np.array_equal(_bits(x["c2"]), _bits(o["dj"]))
    if order == 2 and finalized:
        # di/dj enter the sweep only as the weight area*d (conserve_interp.c:806: (f + gx*di + gy*dj)*area) and
        # cross zero, so the bar "weights within 1e-10 relative" is applied to area*d against its own scale.
        # (d alone is ill-conditioned for sliver cells: clat/area divides two numbers that each lost
        #  ~log10(cell/sliver) digits to cancellation -- in the reference as well.)
        for a, k in ((x["c1"], "di"), (x["c2"], "dj")):
            w, w_ref = a * x["area"], o[k] * o["area"]
            assert np.max(np.abs(w - w_ref)) < RTOL * np.max(np.abs(w_ref)), k
            assert np.max(np.abs(a - o[k])) < 1e-8 * np.max(np.abs(o[k])), k


def make_fields(ni, lon, lat, nz, order, seed=0):
    rng = np.random.default_rng(seed)
    h = 1 if order == 2 else 0
    data, gx, gy = [], [], []
    for t in range(len(lon)):
        data.append(rng.standard_normal((nz, ni + 2 * h, ni + 2 * h)) + 5.0)
        gx.append(rng.standard_normal((nz, ni, ni)))
        gy.append(rng.standard_normal((nz, ni, ni)))
    return data, gx, gy


@pytest.mark.parametrize("order,ni,nlon,nlat", [(1, 48, 180, 90), (2, 48, 144, 90), (2, 96, 360, 180)])
def test_setup_conserve_interp_vs_oracle(fg, gpu_ok, order, ni, nlon, nlat):
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grid_in = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    grid_out = [fg.GridConfig(nlon, nlat, lo, la)]
    interp = [fg.InterpConfig()]
    opcode = fg.CONSERVE_ORDER1 if order == 1 else fg.CONSERVE_ORDER2
    fg.setup_conserve_interp(6, grid_in, 1, grid_out, interp, opcode)
    if ni == 96:
        assert interp[0].nxgrid == 256864                 # BASELINE.md §2 (reference count)
        earth = 4 * np.pi * 6371000.0 ** 2
        assert abs(np.sum(interp[0].area) / earth - 0.999999998703791) < 1e-12   # reference closure, BASELINE.md §2
        # BASELINE config 2 in full: every exchange cell of C96 -> 360x180 against the reference's own code compiled in place
        # (oracle/_ref: six create_xgrid_2dx2d_order2 calls, ~4.5 s of CPU; the bit-identical port when _ref is absent)
        p = fg.XgridPlan.create(2, grid_in, grid_out[0])
        x = p.get_xgrid()                                 # before finalize: c1 / c2 are the reference's xgrid_clon / xgrid_clat
        p.destroy()
        create = orc.ref_create_xgrid if orc.ref_available() else orc.orc_create_xgrid
        off = 0
        for t in range(6):
            r = create(2, ni, ni, nlon, nlat, lon[t], lat[t], lo, la)
            sel = slice(off, off + r["n"])
            assert np.all(x["t_in"][sel] == t)
            for k in ("i_in", "j_in", "i_out", "j_out"):
                assert np.array_equal(x[k][sel], r[k]), (t, k)
                assert np.array_equal(getattr(interp[0], k)[sel], r[k]), (t, k)
            assert np.max(np.abs(x["area"][sel] - r["area"]) / r["area"]) < RTOL
            if orc.host_has_fma():
                for a, b in ((x["area"][sel], r["area"]), (x["c1"][sel], r["clon"]), (x["c2"][sel], r["clat"]), (interp[0].area[sel], r["area"])):
                    assert np.array_equal(_bits(a), _bits(b)), t
            off += r["n"]
        assert off == 256864
        return
    o = orc.orc_setup(order, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    x = dict(t_in=interp[0].t_in, i_in=interp[0].i_in, j_in=interp[0].j_in, i_out=interp[0].i_out,
             j_out=interp[0].j_out, area=interp[0].area, c1=interp[0].di_in, c2=interp[0].dj_in)
    check_xgrid(x, o, order, True)
    expected_total = {(1, 180): 63752, (2, 144): 55904}[(order, nlon)]
    assert interp[0].nxgrid == expected_total
    for t in range(6):
        assert np.max(np.abs(grid_in[t].cell_area - o["cell_area_in"][t]) / o["cell_area_in"][t]) < RTOL


@pytest.mark.parametrize("order,nz,has_missing", [(1, 1, False), (1, 3, False), (2, 1, False), (2, 4, False), (1, 1, True), (2, 1, True),
                                                  (2, 2, False), (2, 8, False), (2, 11, False), (1, 9, False)])
def test_sweep_bitwise_with_oracle_weights(fg, gpu_ok, order, nz, has_missing):
    """Feed the ORACLE's exchange cells to the device sweep: the CSR row order reproduces the reference's
    summation order, so the remapped field is bit-identical (bar: 1e-6 relative)."""
    _sweep_bitwise(fg, order, nz, has_missing, 24, 72, 36)


@pytest.mark.parametrize("order,nz,ni,nlon,nlat", [(2, 3, 32, 6, 3), (1, 2, 64, 4, 2), (2, 8, 64, 4, 2), (1, 8, 64, 4, 2), (1, 7, 48, 24, 12), (2, 1, 64, 4, 2), (1, 1, 32, 6, 3)])
def test_sweep_bitwise_long_rows(fg, gpu_ok, order, nz, ni, nlon, nlat):
    """Fine -> coarse: destination rows of ~400 exchange cells (sorted by a whole wave, k_csr_sort_rows) and of ~3400
    (beyond its LDS staging: the serial path), filled with one atomic per run of equal rows (k_csr_fill).  The row order must
    still be the reference's summation order: bit-identical remapped levels."""
    _sweep_bitwise(fg, order, nz, False, ni, nlon, nlat)


def _sweep_bitwise(fg, order, nz, has_missing, ni, nlon, nlat):
    import torch
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    o = orc.orc_setup(order, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    data, gx, gy = make_fields(ni, lon, lat, nz, order, seed=order * 10 + nz)
    missing = 1.0e20
    gm = None
    if has_missing:
        h = 1 if order == 2 else 0
        for t in range(6):
            m = ((np.add.outer(np.arange(ni + 2 * h), np.arange(ni + 2 * h)) % 10) == 0)
            data[t][0][m] = missing
        gm = [((np.add.outer(np.arange(ni), np.arange(ni)) % 7) == 0).astype(np.int32) for _ in range(6)]
    plan = fg.XgridPlan.create_empty(order, [ni] * 6, [ni] * 6, nlon, nlat)
    plan.set_xgrid(o["t_in"], o["i_in"], o["j_in"], o["i_out"], o["j_out"], o["area"], o.get("di"), o.get("dj"))
    dev = "cuda:0"
    pack = lambda arrs: torch.from_numpy(np.ascontiguousarray(np.stack(
        [np.concatenate([a[k].ravel() for a in arrs]) for k in range(nz)]))).to(dev)
    d_t = pack(data)
    gx_t = pack(gx) if order == 2 else None
    gy_t = pack(gy) if order == 2 else None
    gm_t = torch.from_numpy(np.concatenate([g.ravel() for g in gm])).to(dev) if (gm is not None and order == 2) else None
    out_t = torch.empty(nz * nlon * nlat, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    gs = plan.apply(d_t, out_t, nz=nz, grad_x_t=gx_t, grad_y_t=gy_t, grad_mask_t=gm_t, has_missing=has_missing,
                    missing=missing, want_gsum=True)
    plan.sync()
    out = out_t.cpu().numpy()
    ref, gs_ref = orc.orc_apply(order, o, [ni] * 6, [ni] * 6, [d.reshape(nz, -1) for d in data],
                                [g.reshape(nz, -1) for g in gx] if order == 2 else None,
                                [g.reshape(nz, -1) for g in gy] if order == 2 else None,
                                gm if order == 2 else None, has_missing, missing, nlon, nlat, nz)
    assert np.array_equal(_bits(out), _bits(ref))
    assert abs(gs - gs_ref) <= 1e-12 * abs(gs_ref)
    if nz in (2, 4, 8):                          # the interleaved entry point gives the same bits
        il = lambda t: t.reshape(nz, -1).t().contiguous()
        out_il = torch.empty(nlon * nlat, nz, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        plan.apply_interleaved(nz, il(d_t), out_il, il(gx_t) if order == 2 else None, il(gy_t) if order == 2 else None)
        plan.sync()
        assert np.array_equal(_bits(out_il.t().contiguous().cpu().numpy().ravel()), _bits(ref))
    plan.destroy()


EX_CASES = [
    # order, nz, has_missing, weight, sum, measures, target, monotonic
    (1, 1, False, False, False, False, False, False),
    (2, 3, False, False, False, False, False, False),
    (1, 3, False, True, False, False, False, False),
    (2, 2, False, True, False, False, True, False),
    (1, 1, True, True, True, False, False, False),
    (2, 1, True, False, True, False, False, False),
    (1, 1, True, False, False, True, True, False),
    (2, 1, True, True, False, True, False, False),
    (2, 1, False, False, False, True, True, False),
    (2, 1, False, False, False, False, False, True),
    (2, 1, True, True, False, False, False, True),
    (2, 1, True, False, True, False, False, True),
    (2, 1, False, False, False, True, True, True),
]


@pytest.mark.parametrize("order,nz,has_missing,use_w,use_sum,use_meas,use_target,mono", EX_CASES)
def test_sweep_every_option_bitwise(fg, gpu_ok, order, nz, has_missing, use_w, use_sum, use_meas, use_target, mono):
    """fg_plan_apply_ex against the full-branch oracle (conserve_interp.c:507-910) on the oracle's exchange cells:
    weight field, cell_methods=sum, cell_measures, --target_grid and the monotone limiter, alone and combined.
    The per-entry operation order is the reference's, so results are bit-identical."""
    import torch
    ni, nlon, nlat = 24, 72, 36
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    o = orc.orc_setup(order, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    data, gx, gy = make_fields(ni, lon, lat, nz, order, seed=77 + order)
    if mono:
        gx, gy = [g * 4.0 for g in gx], [g * 4.0 for g in gy]          # make the limiter bite
    rng = np.random.default_rng(3)
    missing = 1.0e20
    h = 1 if order == 2 else 0
    gm = [np.zeros((ni, ni), dtype=np.int32) for _ in range(6)]
    if has_missing:
        for t in range(6):
            m = ((np.add.outer(np.arange(ni + 2 * h), np.arange(ni + 2 * h)) % 10) == 0)
            data[t][0][m] = missing
        gm = [((np.add.outer(np.arange(ni), np.arange(ni)) % 7) == 0).astype(np.int32) for _ in range(6)]
    w = [rng.uniform(0.2, 1.0, (ni, ni)) for _ in range(6)] if use_w else None
    ca = o["cell_area_in"]
    fa = [np.asarray(ca[t]).reshape(ni, ni) * rng.uniform(0.3, 1.0, (ni, ni)) for t in range(6)] if use_meas else None
    cao = orc.orc_get_grid_area(nlon, nlat, lo, la) if use_target else None
    rc, ref, gs_ref = orc.orc_apply_ex(order, o, [ni] * 6, [ni] * 6, [d.reshape(nz, -1) for d in data],
                                       [g.reshape(nz, -1) for g in gx] if order == 2 else None,
                                       [g.reshape(nz, -1) for g in gy] if order == 2 else None,
                                       gm if order == 2 else None, has_missing, missing, nlon, nlat, nz,
                                       weight=w, cell_methods_sum=use_sum, field_area=fa, area_missing=-1e20,
                                       cell_area_in=ca, target_grid=use_target, cell_area_out=cao, monotonic=mono)
    assert rc == 0
    plan = fg.XgridPlan.create_empty(order, [ni] * 6, [ni] * 6, nlon, nlat)
    plan.set_xgrid(o["t_in"], o["i_in"], o["j_in"], o["i_out"], o["j_out"], o["area"], o.get("di"), o.get("dj"))
    dev = "cuda:0"
    pack = lambda arrs: torch.from_numpy(np.ascontiguousarray(np.stack(
        [np.concatenate([a[k].ravel() for a in arrs]) for k in range(nz)]))).to(dev)
    flat = lambda arrs: torch.from_numpy(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in arrs])).to(dev)
    d_t = pack(data)
    gx_t = pack(gx) if order == 2 else None
    gy_t = pack(gy) if order == 2 else None
    gm_t = torch.from_numpy(np.concatenate([g.ravel() for g in gm])).to(dev) if order == 2 else None
    out_t = torch.empty(nz * nlon * nlat, dtype=torch.float64, device=dev)
    w_t = flat(w) if use_w else None
    fa_t = flat(fa) if use_meas else None
    ca_t = flat(ca)
    cao_t = torch.from_numpy(cao).to(dev) if use_target else None
    torch.cuda.synchronize()
    gs = plan.apply_ex(d_t, out_t, nz=nz, grad_x_t=gx_t, grad_y_t=gy_t, grad_mask_t=gm_t, has_missing=has_missing,
                       missing=missing, weight_t=w_t, cell_methods_sum=use_sum, field_area_t=fa_t, area_missing=-1e20,
                       cell_area_in_t=ca_t, cell_area_out_t=cao_t, monotonic=mono, want_gsum=True)
    plan.sync()
    out = out_t.cpu().numpy()
    assert np.array_equal(_bits(out), _bits(ref))
    assert abs(gs - gs_ref) <= 1e-12 * max(abs(gs_ref), 1e-300)
    plan.destroy()


def test_sweep_option_fatal_checks(fg, gpu_ok):
    """The reference's fatal data checks surface as FG_ERR_DATA with the reference's message: a cell_measures area that
    is missing under valid data (conserve_interp.c:584-587)."""
    import torch
    ni, nlon, nlat = 8, 24, 12
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    o = orc.orc_setup(1, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    data = [np.full((1, ni, ni), 2.0) for _ in range(6)]
    fa = [np.asarray(o["cell_area_in"][t]).reshape(ni, ni).copy() for t in range(6)]
    fa[1][3, 3] = -1.0e20
    rc, _, _ = orc.orc_apply_ex(1, o, [ni] * 6, [ni] * 6, [d.reshape(1, -1) for d in data], None, None, None, True, 1e20,
                                nlon, nlat, 1, field_area=fa, area_missing=-1e20, cell_area_in=o["cell_area_in"])
    assert rc == -2
    plan = fg.XgridPlan.create_empty(1, [ni] * 6, [ni] * 6, nlon, nlat)
    plan.set_xgrid(o["t_in"], o["i_in"], o["j_in"], o["i_out"], o["j_out"], o["area"], None, None)
    flat = lambda arrs: torch.from_numpy(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in arrs])).to("cuda:0")
    out_t = torch.empty(nlon * nlat, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    with pytest.raises(fg.FregridHipError, match="data is not missing but area is missing"):
        plan.apply_ex(flat(data), out_t, has_missing=True, missing=1e20, field_area_t=flat(fa), area_missing=-1e20,
                      cell_area_in_t=flat(o["cell_area_in"]))
    with pytest.raises(fg.FregridHipError, match="cell_measures should be false when nz > 1"):
        plan.apply_ex(flat(data), out_t, nz=2, field_area_t=flat(fa), cell_area_in_t=flat(o["cell_area_in"]))
    plan.destroy()


def test_end_to_end_mirror_api_and_conservation(fg, gpu_ok, capsys):
    """setup_conserve_interp + do_scalar_conserve_interp with --check_conserve (the flow of
    tests/fregrid/cubedsphere: C48 -> 144x90, conserve_order2)."""
    ni, nlon, nlat = 48, 144, 90
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grid_in = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    grid_out = [fg.GridConfig(nlon, nlat, lo, la)]
    interp = [fg.InterpConfig()]
    opcode = fg.CONSERVE_ORDER2 | fg.CHECK_CONSERVE
    fg.setup_conserve_interp(6, grid_in, 1, grid_out, interp, opcode)
    import gridutil
    field_in, var = [], fg.VarConfig(name="tvar", interp_method=fg.CONSERVE_ORDER2)
    for t in range(6):
        lc, tc = gridutil.cell_centres(lon[t], lat[t])
        f = 2.0 + 0.1 * gridutil.analytic_field(lc, tc)          # 2 + sin(lon+lat), positive
        field_in.append(fg.FieldConfig(data=np.pad(f, 1, mode="edge")[None], grad_x=(np.cos(lc + tc) * np.ones_like(f))[None],
                                       grad_y=(np.cos(lc + tc))[None], var=[var]))
    field_out = [fg.FieldConfig()]
    gsum_in, gsum_out = fg.do_scalar_conserve_interp(interp, 0, 6, grid_in, 1, grid_out, field_in, field_out, opcode, 1)
    txt = capsys.readouterr().out
    assert "the flux(data*area) sum of tvar: input = " in txt            # conserve_interp.c:904
    assert abs(gsum_out - gsum_in) / abs(gsum_in) < 5e-9                  # the reference's own closure is ~1e-9
    # against the oracle with the oracle's weights: same fields to 1e-6 relative (north_star), in practice ~1e-13
    o = orc.orc_setup(2, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    ref, gs_ref = orc.orc_apply(2, o, [ni] * 6, [ni] * 6, [f.data.reshape(1, -1) for f in field_in],
                                [f.grad_x.reshape(1, -1) for f in field_in], [f.grad_y.reshape(1, -1) for f in field_in],
                                None, False, 0.0, nlon, nlat, 1)
    got = field_out[0].data.ravel()
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-9
    assert abs(gsum_out - gs_ref) / abs(gs_ref) < 1e-11


def _plan_vs_oracle(fg, order, gin, gout, capacity=None):
    grids = [fg.GridConfig(nx, ny, x, y) for (nx, ny, x, y) in gin]
    plan = fg.XgridPlan.create(order, grids, fg.GridConfig(*gout))
    plan.finalize()
    x = plan.get_xgrid()
    o = orc.orc_setup(order, gin, [gout], capacity=capacity)
    check_xgrid(x, o, order, True)
    st = plan.stats()
    plan.destroy()
    return o["n"], st


def test_regional_target_window(fg, gpu_ok):
    """tests/fregrid/remap-file style regional window (15..65N, 230..310E), scaled down."""
    lon, lat = fg.gnomonic_ed_corners(32)
    lo, la = fg.latlon_corners(64, 40, 230.0, 310.0, 15.0, 65.0)
    n, _ = _plan_vs_oracle(fg, 2, [(32, 32, lon[t], lat[t]) for t in range(6)], (64, 40, lo, la))
    assert n > 0


def test_curvilinear_target_with_pole_caps(fg, gpu_ok):
    """lat-lon source -> cubed-sphere tiles as targets (2dx2d, neither side lat-lon for the clip): tile 3 has a
    pole vertex (5-vertex cells, wide longitude boxes -> wide lists and the general clip kernel)."""
    lon, lat = fg.gnomonic_ed_corners(16)
    lo, la = fg.latlon_corners(90, 45)
    for t in (0, 2, 5):
        n, st = _plan_vs_oracle(fg, 2, [(90, 45, lo, la)], (16, 16, lon[t], lat[t]))
        assert n > 0
        if t == 2:
            assert st["deferred"] > 0
    # cubed sphere -> cubed sphere of another resolution, polar tiles
    lon2, lat2 = fg.gnomonic_ed_corners(10)
    n, st = _plan_vs_oracle(fg, 1, [(16, 16, lon[2], lat[2]), (16, 16, lon[5], lat[5])], (10, 10, lon2[2], lat2[2]))
    assert n > 0


def test_polar_tile_targets_with_long_cells_in_the_bins(fg, gpu_ok):
    """1-degree lat-lon source -> the two polar tiles of C48: the target's cells around the pole cover many bin columns (rings) or
    three and more bin rows (the tile's diagonals) and are stored as several copies in the bins, of which a query takes the first
    one its window meets (d_copy_first in xgrid_kernels.hip).  Every exchange cell against the oracle, both orders; the source
    window that straddles the 0/360 seam and the rows next to the poles are part of the global source grid."""
    lon, lat = fg.gnomonic_ed_corners(48)
    lo, la = fg.latlon_corners(360, 180)
    for t, order in ((2, 1), (5, 2)):
        n, st = _plan_vs_oracle(fg, order, [(360, 180, lo, la)], (48, 48, lon[t], lat[t]))
        assert n > 48 * 48
    lon96, lat96 = fg.gnomonic_ed_corners(96)            # finer on both sides: more copies per ring cell
    lo5, la5 = fg.latlon_corners(720, 360)
    n, st = _plan_vs_oracle(fg, 2, [(720, 360, lo5, la5)], (96, 96, lon96[2], lat96[2]))
    assert n > 96 * 96
    # a source whose longitudes start at -280 (the tripolar convention) meets the copies through the +-2 pi shifts
    lo2, la2 = fg.latlon_corners(180, 90, -280.0, 80.0, -90.0, 90.0)
    n, st = _plan_vs_oracle(fg, 1, [(180, 90, lo2, la2)], (48, 48, lon[2], lat[2]))
    assert n > 48 * 48


def test_tripolar_both_directions(fg, gpu_ok):
    """BASELINE config 5 scaled down: tripolar ocean grid (longitudes -280..80, bipolar Arctic cap with the fold
    along the top row) <-> cubed-sphere tiles, neither side lat-lon.  The generator is unpinned input synthesis
    (grid_gen.c); parity of the search on it is against the oracle."""
    tlon, tlat = fg.tripolar_corners(90, 54)
    lon, lat = fg.gnomonic_ed_corners(16)
    tot = 0
    for t in range(6):                                   # ocean -> atmosphere tile t
        n, st = _plan_vs_oracle(fg, 2 if t in (2, 4) else 1, [(90, 54, tlon, tlat)], (16, 16, lon[t], lat[t]))
        tot += n
    assert tot > 6 * 256
    # atmosphere (all six tiles) -> ocean, second order
    n, st = _plan_vs_oracle(fg, 2, [(16, 16, lon[t], lat[t]) for t in range(6)], (90, 54, tlon, tlat))
    assert n > 90 * 54
    # exchange-grid area closes on the ocean grid's own area (every ocean cell is fully covered by the sphere)
    grids = [fg.GridConfig(16, 16, lon[t], lat[t]) for t in range(6)]
    plan = fg.XgridPlan.create(1, grids, fg.GridConfig(90, 54, tlon, tlat))
    plan.finalize()
    x = plan.get_xgrid()
    plan.destroy()
    got = np.bincount(x["j_out"].astype(np.int64) * 90 + x["i_out"], weights=x["area"], minlength=90 * 54)
    area = orc.orc_get_grid_area(90, 54, tlon, tlat)
    assert np.max(np.abs(got - area) / area) < 2e-3      # poly_area's great-circle-vs-parallel edge model, not a bug


def test_six_output_tiles_accumulate_in_the_references_order(fg, gpu_ok):
    """Order 2 onto a cubed sphere: a source cell that meets two output tiles gets its (area, clon, clat) sums from both, and
    the reference adds the exchange cells of output tile 0, then of tile 1, ... onto ONE accumulator (conserve_interp.c:136-147
    sit outside the n loop, :216-221).  fg_plan_accumulate_cell_sums continues the running total from plan to plan, so di / dj
    carry the oracle's bits -- a sum of per-tile partial sums would not."""
    no = 12
    lo, la = fg.latlon_corners(60, 30)
    clon, clat = fg.gnomonic_ed_corners(no)
    grid_in = [fg.GridConfig(60, 30, lo, la)]
    grid_out = [fg.GridConfig(no, no, clon[t], clat[t]) for t in range(6)]
    interp = [fg.InterpConfig() for _ in range(6)]
    fg.setup_conserve_interp(1, grid_in, 6, grid_out, interp, fg.CONSERVE_ORDER2)
    o = orc.orc_setup(2, [(60, 30, lo, la)], [(no, no, clon[t], clat[t]) for t in range(6)])
    shared = 0
    for n in range(6):
        ic = interp[n]
        sl = slice(int(o["xoff"][n]), int(o["xoff"][n + 1]))
        assert ic.nxgrid == sl.stop - sl.start > 0
        for k, ok in (("i_in", "i_in"), ("j_in", "j_in"), ("i_out", "i_out"), ("j_out", "j_out")):
            assert np.array_equal(getattr(ic, k), o[ok][sl]), (n, k)
        if orc.host_has_fma():
            assert np.array_equal(_bits(ic.area), _bits(o["area"][sl]))
            assert np.array_equal(_bits(ic.di_in), _bits(o["di"][sl])) and np.array_equal(_bits(ic.dj_in), _bits(o["dj"][sl])), n
        else:
            assert np.allclose(ic.di_in, o["di"][sl], rtol=1e-9, atol=1e-12)
        ic.plan.destroy()
    src = o["j_in"].astype(np.int64) * 60 + o["i_in"]
    tile = np.searchsorted(o["xoff"][1:], np.arange(o["n"]), side="right")
    for c in np.unique(src):
        shared += len(np.unique(tile[src == c])) > 1
    assert shared > 100                                   # many source cells do meet two or three output tiles


def test_coarse_to_fine_and_fine_to_coarse(fg, gpu_ok):
    lon, lat = fg.gnomonic_ed_corners(8)
    lo, la = fg.latlon_corners(240, 120)
    n1, st = _plan_vs_oracle(fg, 2, [(8, 8, lon[t], lat[t]) for t in (0, 2)], (240, 120, lo, la))
    assert st["heavy"] > 0                      # every source cell sees hundreds of targets
    lon, lat = fg.gnomonic_ed_corners(64)
    lo, la = fg.latlon_corners(12, 6)
    n2, _ = _plan_vs_oracle(fg, 2, [(64, 64, lon[t], lat[t]) for t in (1, 5)], (12, 6, lo, la))
    assert n1 > 0 and n2 > 0


def test_fast_search_overflow_falls_back_to_exact_mode(fg, gpu_ok):
    """The single-sync search sizes its pair buffers at 8*max(nsrc, ndst)+65536; 300 meridional strips against 300 zonal
    bands make 90 000 pairs from 300 + 300 cells, so the attempt overflows and is repeated with exact sizes -- same
    exchange cells either way, and equal to forcing exact mode from the start."""
    lo1, la1 = fg.latlon_corners(300, 1, 0.0, 170.0, -60.0, 60.0)       # cells narrower than pi: fix_lon folds wider ones
    lo2, la2 = fg.latlon_corners(1, 300, 10.0, 160.0, -50.0, 50.0)
    gin, gout = [(300, 1, lo1, la1)], (1, 300, lo2, la2)
    n, st = _plan_vs_oracle(fg, 2, gin, gout, capacity=200000)
    assert n > 70000 and st["exact_mode"] == 1 and st["pairs"] >= n
    # an ordinary case stays on the fast path; forcing exact mode gives the same plan
    lon, lat = fg.gnomonic_ed_corners(16)
    lo, la = fg.latlon_corners(48, 24)
    gin, gout = [(16, 16, lon[t], lat[t]) for t in range(6)], (48, 24, lo, la)
    n1, st1 = _plan_vs_oracle(fg, 2, gin, gout)
    assert st1["exact_mode"] == 0
    fg.lib().fg_set_search_mode(1)
    try:
        n2, st2 = _plan_vs_oracle(fg, 2, gin, gout)
    finally:
        fg.lib().fg_set_search_mode(0)
    assert st2["exact_mode"] == 1 and n1 == n2 and st1["pairs"] == st2["pairs"]


@pytest.fixture
def generic_path(fg):
    """the machinery of the generic (bins) search on lat-lon targets, which the rectilinear path would otherwise take"""
    fg.lib().fg_set_search_rect(0)
    yield
    fg.lib().fg_set_search_rect(1)


def test_chunked_search_gives_the_same_plan(fg, gpu_ok, generic_path):
    """Large grids are searched in chunks of source cells (clip of one chunk on a second stream beside the candidate scan of
    the next): force 1, 3 and 8 chunks on a small case -- exchange cells, order-2 integrals and the sweep must not change."""
    lon, lat = fg.gnomonic_ed_corners(24)
    lo, la = fg.latlon_corners(72, 36)
    gin, gout = [(24, 24, lon[t], lat[t]) for t in range(6)], (72, 36, lo, la)
    res = []
    try:
        for k in (1, 3, 8):
            fg.lib().fg_set_search_chunks(k)
            res.append(_plan_vs_oracle(fg, 2, gin, gout))
    finally:
        fg.lib().fg_set_search_chunks(0)
    assert res[0][0] == res[1][0] == res[2][0] > 0
    assert res[0][1]["pairs"] == res[1][1]["pairs"] == res[2][1]["pairs"]


def test_source_cell_culling_keeps_the_plan(fg, gpu_ok):
    """fg_set_search_cull(1): a rank that owns one latitude band builds records only for the source cells that can meet it.
    Same exchange cells, order-2 integrals and cell sums as without culling, on a polar and on a mid-latitude band."""
    ni, nlon, nlat = 24, 72, 36
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    gin = [(ni, ni, lon[t], lat[t]) for t in range(6)]
    for (j0, j1) in ((0, 5), (14, 23), (30, 36)):
        band = (nlon, j1 - j0, lo[j0:j1 + 1], la[j0:j1 + 1])
        res = []
        try:
            for cull in (0, 1):
                fg.lib().fg_set_search_cull(cull)
                plan = fg.XgridPlan.create(2, [fg.GridConfig(*g) for g in gin], fg.GridConfig(*band))
                a_in, _ = plan.get_cell_area(band[0] * band[1])
                plan.finalize()
                x = plan.get_xgrid()
                res.append((plan.nxgrid, x, a_in))
                plan.destroy()
        finally:
            fg.lib().fg_set_search_cull(0)
        (n0, x0, a0), (n1, x1, a1) = res
        assert n0 == n1 > 0
        for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
            assert np.array_equal(x0[k], x1[k])
        for k in ("area", "c1", "c2"):
            assert np.array_equal(x0[k].view(np.uint64), x1[k].view(np.uint64))
        live = a1 != 0
        assert 0 < live.sum() < live.size and np.array_equal(a0[live], a1[live])      # culled cells report area 0
        s = x0["t_in"].astype(np.int64) * ni * ni + x0["j_in"].astype(np.int64) * ni + x0["i_in"]
        assert live[s].all()                                                            # every cell with exchange cells was kept


def test_culled_plan_sweeps_like_the_unculled_one(fg, gpu_ok):
    """A culled plan must still sweep: the order-2 sweep merges field and gradients for EVERY source cell, so the field index of
    the culled cells has to be there too (a block of source cells wholly outside the band leaves the record kernel early; the
    2-rank bench rehearsal faulted on its uninitialised indices).  C96 against a narrow band: most 256-cell blocks are culled
    whole.  The pool is poisoned first so that forgotten stores show up as wild indices rather than as lucky zeros."""
    import torch
    ni, nlon, nlat, nz = 96, 360, 180, 8
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    rng = np.random.default_rng(3)
    src = torch.from_numpy(rng.standard_normal((nz, 6 * (ni + 2) ** 2))).to("cuda:0")
    gx = torch.from_numpy(rng.standard_normal((nz, 6 * ni * ni))).to("cuda:0")
    gy = torch.from_numpy(rng.standard_normal((nz, 6 * ni * ni))).to("cuda:0")
    for (j0, j1) in ((100, 112), (0, 9)):
        band = fg.GridConfig(nlon, j1 - j0, np.ascontiguousarray(lo[j0:j1 + 1]), np.ascontiguousarray(la[j0:j1 + 1]))
        outs = []
        try:
            for cull in (0, 1):
                # poison: blocks the pool hands out next hold 0x7f7f... (int 2139062143, a NaN-ish double)
                for nbytes in (6 * ni * ni * 4 + 4, 6 * ni * ni * 8, 3 * 6 * ni * ni * 8):
                    pz = fg.lib().fg_dev_alloc(nbytes, 0)
                    junk = np.full(nbytes, 0x7f, dtype=np.uint8)
                    assert fg.lib().fg_dev_upload(C.c_void_p(pz), junk.ctypes.data_as(C.c_void_p), nbytes) == 0
                    fg.lib().fg_dev_free(C.c_void_p(pz))
                fg.lib().fg_set_search_cull(cull)
                plan = fg.XgridPlan.create(2, grids, band)
                plan.finalize()
                out = torch.full((nz, nlon * (j1 - j0)), np.nan, dtype=torch.float64, device="cuda:0")
                torch.cuda.synchronize()                   # (the fill runs on torch's stream, the library on the plan's own)
                plan.apply(src, out, nz=nz, grad_x_t=gx, grad_y_t=gy); plan.sync()
                outs.append(out.cpu().numpy())
                plan.destroy()
        finally:
            fg.lib().fg_set_search_cull(0)
        assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])


def test_plan_trim_keeps_the_plan(fg, gpu_ok):
    """fg_plan_trim re-allocates the capacity-sized exchange-cell arrays at nxgrid entries: before or after
    fg_plan_finalize, the exchange cells and the sweep are unchanged."""
    import torch
    ni, nlon, nlat = 16, 48, 24
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    src = torch.from_numpy(np.random.default_rng(5).standard_normal((3, 6 * ni * ni))).to("cuda:0")
    outs, xs = [], []
    for when in ("never", "before", "after"):
        p = fg.XgridPlan.create(1, grids, fg.GridConfig(nlon, nlat, lo, la))
        if when == "before":
            assert fg.lib().fg_plan_trim(p._h) == 0
        p.finalize()
        if when == "after":
            assert fg.lib().fg_plan_trim(p._h) == 0
        out = torch.empty(3, nlon * nlat, dtype=torch.float64, device="cuda:0")
        p.apply(src, out, nz=3); p.sync()
        outs.append(out.cpu().numpy()); xs.append(p.get_xgrid())
        p.destroy()
    for o, x in zip(outs[1:], xs[1:]):
        assert np.array_equal(o, outs[0])
        for k in ("t_in", "i_in", "j_in", "i_out", "j_out", "area"):
            assert np.array_equal(x[k], xs[0][k])


def test_sweep_tile_mapping_keeps_the_bits(fg, gpu_ok):
    """fg_set_apply_xcd: identity, one band per XCD, and chunked tile -> XCD mappings (chunk sizes that divide the grid and
    that leave a tail) are permutations of the tiles: the sweep's output is bitwise the same for each."""
    import torch
    ni, nlon, nlat = 48, 360, 180
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    p = fg.XgridPlan.create(1, grids, fg.GridConfig(nlon, nlat, lo, la))
    p.finalize()
    src = torch.from_numpy(np.random.default_rng(11).standard_normal((8, 6 * ni * ni))).to("cuda:0")
    outs = []
    try:
        for mode in (0, 1, 2, 3, 7, 64, 1000):
            fg.lib().fg_set_apply_xcd(mode)
            out = torch.full((8, nlon * nlat), np.nan, dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()                   # (the fill runs on torch's stream, the library on the plan's own)
            p.apply(src, out, nz=8); p.sync()
            outs.append(out.cpu().numpy())
    finally:
        fg.lib().fg_set_apply_xcd(64)
        p.destroy()
    assert np.isfinite(outs[0]).all()
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("ni,nlon,nlat", [(48, 360, 180), (96, 90, 45), (24, 1440, 720), (96, 360, -1)])
def test_entry_parallel_sweep_keeps_the_bits(fg, gpu_ok, ni, nlon, nlat):
    """fg_set_apply_ep: the entry-parallel 8-level order-2 sweep (products through LDS, row sums in CSR order) against the
    row-serial kernel, on similar cells, on fine -> coarse (rows of ~20 exchange cells: the host keeps the row-serial kernel,
    or a tile overflows the product table and takes the row-serial loop inside the kernel) and on coarse -> fine."""
    import torch
    lon, lat = fg.gnomonic_ed_corners(ni)
    if nlat > 0:
        lo, la = fg.latlon_corners(nlon, nlat)
    else:
        # tall cells in the south (12 exchange cells per row), short ones in the north (3): 4.3 on average, so the entry-parallel
        # kernel is chosen and its southern tiles overflow the product table
        edges = np.deg2rad(np.concatenate([np.linspace(-90.0, 0.0, 20), np.linspace(0.0, 90.0, 161)[1:]]))
        nlat = edges.size - 1
        lo, la = np.meshgrid(np.linspace(0.0, 2 * np.pi, nlon + 1), edges)
        lo, la = np.ascontiguousarray(lo), np.ascontiguousarray(la)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    p = fg.XgridPlan.create(2, grids, fg.GridConfig(nlon, nlat, lo, la))
    p.finalize()
    if nlon == 360 and ni == 96:
        assert p.nxgrid <= 6 * nlon * nlat and p.nxgrid > 3.5 * nlon * nlat
    rng = np.random.default_rng(12)
    nc = 6 * ni * ni
    src = torch.from_numpy(rng.standard_normal((8, 6 * (ni + 2) ** 2))).to("cuda:0")
    gx = torch.from_numpy(rng.standard_normal((8, nc))).to("cuda:0")
    gy = torch.from_numpy(rng.standard_normal((8, nc))).to("cuda:0")
    outs = []
    try:
        for ep in (0, 1):
            fg.lib().fg_set_apply_ep(ep)
            out = torch.full((8, nlon * nlat), np.nan, dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()                   # (the fill runs on torch's stream, the library on the plan's own)
            p.apply(src, out, nz=8, grad_x_t=gx, grad_y_t=gy); p.sync()
            outs.append(out.cpu().numpy())
    finally:
        fg.lib().fg_set_apply_ep(1)
        p.destroy()
    assert np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1])


def test_bin_record_overflow_falls_back_to_exact_mode(fg, gpu_ok, generic_path):
    """Bins no larger than the target cells (the caller's mean cell size is an input): every target cell then spans four bin
    rows, lands in the per-row wide lists four times and the single-sync search's record buffer (3 per target cell) is too
    small -- the kernels must stay inside it and the search must be repeated with exact sizes, giving the oracle's list."""
    import torch
    ni, nlon, nlat = 16, 96, 48
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    dev = "cuda:0"
    lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]
    lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
    lo_t, la_t = torch.from_numpy(lo).to(dev), torch.from_numpy(la).to(dev)
    torch.cuda.synchronize()
    # choose_bins makes bins 1.25 x the mean cell: pass means that make them exactly one target cell
    plan = fg.XgridPlan.create_dev(1, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t,
                                   (np.pi / nlat) / 2.5, (2 * np.pi / nlon) / 1.25)
    st = plan.stats()
    assert st["exact_mode"] == 1 and st["bin_entries"] > 3 * nlon * nlat + 4096
    x = plan.get_xgrid()
    plan.destroy()
    o = orc.orc_setup(1, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    check_xgrid(x, o, 1, False)


def test_bad_inputs_are_reported_not_crashed(fg, gpu_ok):
    """Corner arrays in degrees (a common caller mistake) or with NaNs: the plan call returns FG_ERR_ARG with a message; the
    kernels that already ran on the garbage stay inside their tables."""
    lon, lat = fg.gnomonic_ed_corners(8)
    lo, la = fg.latlon_corners(24, 12)
    grids = [fg.GridConfig(8, 8, np.degrees(lon[0]), np.degrees(lat[0]))]
    with pytest.raises(fg.FregridHipError, match="radians expected"):
        fg.XgridPlan.create(2, grids, fg.GridConfig(24, 12, lo, la))
    bad = la.copy(); bad[3, 5] = np.nan
    with pytest.raises(fg.FregridHipError, match="radians expected"):
        fg.XgridPlan.create(1, [fg.GridConfig(8, 8, lon[0], lat[0])], fg.GridConfig(24, 12, lo, bad))
    # and the library is still healthy afterwards
    n, _ = _plan_vs_oracle(fg, 1, [(8, 8, lon[0], lat[0])], (24, 12, lo, la))
    assert n > 0


def test_degenerate_sizes_and_empty_overlap(fg, gpu_ok):
    lo1, la1 = fg.latlon_corners(1, 1, 10.0, 20.0, 10.0, 20.0)
    lo2, la2 = fg.latlon_corners(1, 1, 15.0, 30.0, 5.0, 15.0)
    n, _ = _plan_vs_oracle(fg, 2, [(1, 1, lo1, la1)], (1, 1, lo2, la2))
    assert n == 1
    lo3, la3 = fg.latlon_corners(4, 4, 100.0, 120.0, -40.0, -20.0)
    plan = fg.XgridPlan.create(1, [fg.GridConfig(1, 1, lo1, la1)], fg.GridConfig(4, 4, lo3, la3))
    assert plan.nxgrid == 0
    plan.finalize()
    import torch
    d = torch.ones(1, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()                   # (filled on torch's stream; the library works on the plan's own)
    out = torch.empty(16, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    plan.apply(d, out, nz=1)
    plan.sync()
    assert np.all(out.cpu().numpy() == -1.0e20)         # untouched cells -> missing (= -MAXVAL), conserve_interp.c:838
    plan.destroy()
    # touching but not overlapping boxes: the reference's strict rejects drop them
    lo4, la4 = fg.latlon_corners(1, 1, 20.0, 30.0, 10.0, 20.0)
    plan = fg.XgridPlan.create(1, [fg.GridConfig(1, 1, lo1, la1)], fg.GridConfig(1, 1, lo4, la4))
    assert plan.nxgrid == 0
    plan.destroy()


def test_conserve_interp_b1(fg, gpu_ok):
    """interp.c:262 drop-in."""
    lo1, la1 = fg.latlon_corners(36, 18)
    lo2, la2 = fg.latlon_corners(20, 10)
    rng = np.random.default_rng(3)
    data = rng.standard_normal(36 * 18)
    got = fg.conserve_interp(36, 18, 20, 10, lo1, la1, lo2, la2, None, data)
    x = orc.orc_create_xgrid(1, 36, 18, 20, 10, lo1, la1, lo2, la2)
    dst_area = np.zeros(200)
    for n in range(x["n"]):
        dst_area[x["j_out"][n] * 20 + x["i_out"][n]] += x["area"][n]
    exp = np.zeros(200)
    for n in range(x["n"]):
        d = x["j_out"][n] * 20 + x["i_out"][n]
        exp[d] += data[x["j_in"][n] * 36 + x["i_in"][n]] * (x["area"][n] / dst_area[d])
    assert np.max(np.abs(got - exp)) < 1e-12


def test_polygon_known_answers_on_device(fg, gpu_ok):
    """Cases 15..26 of the reference's embedded test main through the device clip (clip_2dx2d, fix_lon,
    poly_area behind their libfrencutils names): vertices bit-exact, areas 1e-10."""
    L = fg.lib()
    dp = C.POINTER(C.c_double)
    P = lambda v: v.ctypes.data_as(dp)
    cases = json.load(open(os.path.join(GOLD, "clip_cases.json")))["cases"]
    for c in cases:
        pad = lambda v: np.array(list(v) + [0.0] * (30 - len(v)))
        x1, y1 = pad(np.array(c["lon1_deg"]) * D2R), pad(np.array(c["lat1_deg"]) * D2R)
        x2, y2 = pad(np.array(c["lon2_deg"]) * D2R), pad(np.array(c["lat2_deg"]) * D2R)
        xo, yo = np.zeros(30), np.zeros(30)
        n = L.clip_2dx2d(P(x1), P(y1), len(c["lon1_deg"]), P(x2), P(y2), len(c["lon2_deg"]), P(xo), P(yo))
        ref = c["ref"]
        assert n == ref["n_clip"], c["case"]
        assert np.array_equal(_bits(xo[:n]), _bits(np.array(ref["clip_lon"])))
        assert np.array_equal(_bits(yo[:n]), _bits(np.array(ref["clip_lat"])))
        if n > 0 and n <= 8:
            nf = L.fix_lon(P(xo), P(yo), n, np.pi)
            assert nf == ref["n_out_fixed"]
            assert np.array_equal(_bits(xo[:nf]), _bits(np.array(ref["out_lon"])))
            a = L.poly_area(P(xo), P(yo), nf)
            assert abs(a - ref["area_out"]) <= RTOL * ref["area_out"]
    # ctrlon / ctrlat on a C48 polar cell against the oracle
    lon, lat = fg.gnomonic_ed_corners(48)
    O = orc.oracle()
    x = np.array([lon[2, 24, 24], lon[2, 24, 25], lon[2, 25, 25], lon[2, 25, 24]] + [0.0] * 20)
    y = np.array([lat[2, 24, 24], lat[2, 24, 25], lat[2, 25, 25], lat[2, 25, 24]] + [0.0] * 20)
    xa, ya = x.copy(), y.copy()
    n = L.fix_lon(P(x), P(y), 4, np.pi)
    assert n == O.orc_fix_lon(P(xa), P(ya), 4, np.pi) == 5
    assert np.array_equal(_bits(x[:n]), _bits(xa[:n]))
    cl = float(np.mean(x[:n]))
    for f_dev, f_orc, args in ((L.poly_ctrlon, O.orc_poly_ctrlon, (cl,)), (L.poly_ctrlat, O.orc_poly_ctrlat, ())):
        a, b = f_dev(P(x), P(y), n, *args), f_orc(P(xa), P(ya), n, *args)
        assert abs(a - b) <= 1e-10 * max(abs(b), 1e6)


def test_full_size_properties_c384(fg, gpu_ok):
    """BASELINE.json's headline configuration (C384 -> 1440x720, order 2) at full size, checked through
    size-independent properties (the brute-force oracle would need ~15 CPU-minutes here): the reference's
    own counts, canonical order, area closure, partition of unity, linearity and
    constant preservation of the sweep, and conservation."""
    import torch
    ni, nlon, nlat = 384, 1440, 720
    counts = json.load(open(os.path.join(GOLD, "counts.json")))["C384->1440x720 o2"]
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    plan = fg.XgridPlan.create(2, grids, fg.GridConfig(nlon, nlat, lo, la))
    assert plan.nxgrid == counts["total"] == 4160000
    st = plan.stats()
    assert st["borderline"] == 0                      # no pair within 1e-9 (relative) of the 1e-6 area threshold
    plan.finalize()
    x = plan.get_xgrid()
    per_tile = np.bincount(x["t_in"], minlength=6)
    assert list(per_tile) == [counts["tiles_1245"], counts["tiles_1245"], counts["tiles_36"], counts["tiles_1245"],
                              counts["tiles_1245"], counts["tiles_36"]]
    assert [int(x[k][0]) for k in ("i_in", "j_in", "i_out", "j_out")] == counts["first_xcell_tile1"]
    assert abs(x["area"][0] - counts["first_area_tile1"]) < 1e-10 * counts["first_area_tile1"]
    s = x["t_in"].astype(np.int64) * ni * ni + x["j_in"].astype(np.int64) * ni + x["i_in"]
    d = x["j_out"].astype(np.int64) * nlon + x["i_out"]
    assert np.all(np.diff(s * (nlon * nlat) + d) > 0)  # canonical order, no duplicates
    earth = 4 * np.pi * 6371000.0 ** 2
    assert abs(np.sum(x["area"]) / earth - 0.999999999040900) < 1e-12          # BASELINE.md §2, reference closure
    a_in, a_out = plan.get_cell_area(nlon * nlat)
    cov = np.bincount(d, weights=x["area"], minlength=nlon * nlat)
    assert np.max(np.abs(cov - a_out) / a_out) < 1e-4                          # --check_conserve bound, conserve_interp.c:479
    for arr in (x["c1"], x["c2"]):                                             # centroid distances balance per source cell
        tot = np.bincount(s, weights=arr * x["area"], minlength=6 * ni * ni)
        assert np.max(np.abs(tot)) < 1e-6 * np.max(a_in) * 1e-3
    dev = "cuda:0"
    F, ncell = 6 * (ni + 2) ** 2, 6 * ni * ni
    g = torch.Generator(device="cpu").manual_seed(1)
    f1 = torch.randn(F, dtype=torch.float64, generator=g); f2 = torch.randn(F, dtype=torch.float64, generator=g)
    gx = torch.randn(ncell, dtype=torch.float64, generator=g); gy = torch.randn(ncell, dtype=torch.float64, generator=g)
    data = torch.stack([f1, f2, 2.0 * f1 - 3.0 * f2, torch.full((F,), 7.5, dtype=torch.float64)]).to(dev)
    gxs = torch.stack([gx, gy, 2.0 * gx - 3.0 * gy, torch.zeros(ncell, dtype=torch.float64)]).to(dev)
    gys = torch.stack([gy, gx, 2.0 * gy - 3.0 * gx, torch.zeros(ncell, dtype=torch.float64)]).to(dev)
    out = torch.empty(4, nlon * nlat, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    plan.apply(data, out, nz=4, grad_x_t=gxs, grad_y_t=gys)
    plan.sync()
    o = out.cpu().numpy()
    assert np.max(np.abs(o[2] - (2.0 * o[0] - 3.0 * o[1]))) < 1e-11           # linearity
    assert np.max(np.abs(o[3] - 7.5)) < 1e-14                                 # constants are preserved
    # conservation with a positive field and zero gradient: sum(out*covered area) == sum(f*xarea)
    pos = torch.full((1, F), 1.0, dtype=torch.float64, device=dev)
    z = torch.zeros(1, ncell, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()                   # (filled on torch's stream; the library works on the plan's own)
    o1 = torch.empty(nlon * nlat, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    gs = plan.apply(pos, o1, nz=1, grad_x_t=z, grad_y_t=z, want_gsum=True)
    assert abs(gs - np.sum(x["area"])) < 1e-12 * gs
    plan.destroy()


def test_remap_file_write_then_read_branch(fg, gpu_ok, tmp_path):
    """BASELINE configs 4/5: 'write remap_file' then 'read cached remap_file' (set_remap_file semantics,
    fregrid_util.c:1946-1992): the READ branch must sweep exactly like the plan that computed the weights."""
    import torch
    ni, nlon, nlat = 32, 96, 48
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grid_in = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    path = str(tmp_path / "remap.tile1.nc")
    interp = [fg.InterpConfig(remap_file=path)]
    fg.setup_conserve_interp(6, grid_in, 1, [fg.GridConfig(nlon, nlat, lo, la)], interp, fg.CONSERVE_ORDER2 | fg.WRITE)
    assert fg.lib().fg_remap_read_size(path.encode()) == interp[0].nxgrid
    interp2 = [fg.InterpConfig(remap_file=path, file_exist=1)]
    fg.setup_conserve_interp(6, grid_in, 1, [fg.GridConfig(nlon, nlat, lo, la)], interp2, fg.CONSERVE_ORDER2 | fg.READ)
    assert interp2[0].nxgrid == interp[0].nxgrid
    for k in ("t_in", "i_in", "j_in", "i_out", "j_out", "di_in", "dj_in"):
        assert np.array_equal(getattr(interp2[0], k), getattr(interp[0], k)), k
    dev = "cuda:0"
    g = torch.Generator().manual_seed(4)
    data = torch.randn(1, 6 * (ni + 2) ** 2, dtype=torch.float64, generator=g).to(dev)
    gx = torch.randn(1, 6 * ni * ni, dtype=torch.float64, generator=g).to(dev)
    gy = torch.randn(1, 6 * ni * ni, dtype=torch.float64, generator=g).to(dev)
    o1 = torch.empty(nlon * nlat, dtype=torch.float64, device=dev); o2 = torch.empty_like(o1)
    torch.cuda.synchronize()
    interp[0].plan.apply(data, o1, nz=1, grad_x_t=gx, grad_y_t=gy); interp[0].plan.sync()
    interp2[0].plan.apply(data, o2, nz=1, grad_x_t=gx, grad_y_t=gy); interp2[0].plan.sync()
    a, b = o1.cpu().numpy(), o2.cpu().numpy()
    # the file round trip rescales the areas (area/garea*garea, read_mosaic.c:432 + conserve_interp.c:86): last-bit changes only
    assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(a))
    # rank-local read: keep only the cells of a latitude band (conserve_interp.c:93-112)
    j0, j1 = fg.band_rows(nlat, 2, 1)
    band = fg.GridConfig(nlon, j1 - j0, lo[j0:j1 + 1], la[j0:j1 + 1], jsc=j0)
    interp3 = [fg.InterpConfig(remap_file=path, file_exist=1)]
    fg.setup_conserve_interp(6, grid_in, 1, [band], interp3, fg.CONSERVE_ORDER2 | fg.READ)
    o3 = torch.empty(nlon * (j1 - j0), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    interp3[0].plan.apply(data, o3, nz=1, grad_x_t=gx, grad_y_t=gy); interp3[0].plan.sync()
    assert np.array_equal(o3.cpu().numpy(), b.reshape(nlat, nlon)[j0:j1].ravel())


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("ni,nlon,nlat", [(48, 144, 90), (96, 90, 45), (96, 24, 12), (48, 6, 3)])
def test_single_level_entry_parallel_sweep_keeps_the_bits(fg, gpu_ok, order, ni, nlon, nlat):
    """k_apply_ep1 (what fregrid's level loop calls: one level per do_scalar_conserve_interp) against the lane-per-row kernel
    k_apply1 (fg_set_apply_ep(0)), with and without missing values / gradient mask, on rows of ~4, ~20, ~300 and ~2000 exchange
    cells (the last two walk a row in several chunks of the product table); the flux sum too."""
    import torch
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    p = fg.XgridPlan.create(order, grids, fg.GridConfig(nlon, nlat, lo, la))
    p.finalize()
    nc = 6 * ni * ni
    nf = 6 * (ni + 2) ** 2 if order == 2 else nc
    rng = np.random.default_rng(21)
    missing = -1.0e10
    src = rng.standard_normal(nf)
    src_m = src.copy(); src_m[rng.random(nf) < 0.2] = missing
    gx, gy = rng.standard_normal(nc), rng.standard_normal(nc)
    gm = (rng.random(nc) < 0.3).astype(np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") if a is not None else None
    res = {}
    try:
        for ep in (0, 1):
            fg.lib().fg_set_apply_ep(ep)
            for has_missing in (False, True):
                out = torch.full((nlon * nlat,), np.nan, dtype=torch.float64, device="cuda:0")
                torch.cuda.synchronize()                   # (the fill runs on torch's stream, the library on the plan's own)
                g = p.apply(t(src_m if has_missing else src), out, nz=1, grad_x_t=t(gx) if order == 2 else None,
                            grad_y_t=t(gy) if order == 2 else None, grad_mask_t=t(gm) if (order == 2 and has_missing) else None,
                            has_missing=has_missing, missing=missing, want_gsum=True)
                p.sync()
                res[(ep, has_missing)] = (out.cpu().numpy(), g)
    finally:
        fg.lib().fg_set_apply_ep(1)
        p.destroy()
    for has_missing in (False, True):
        a, ga = res[(0, has_missing)]; b, gb = res[(1, has_missing)]
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), has_missing
        assert np.float64(ga).view(np.uint64) == np.float64(gb).view(np.uint64), has_missing
    assert (res[(1, True)][0] == missing).sum() >= 0 and np.isfinite(res[(1, False)][0]).all()


@pytest.mark.parametrize("order,mono", [(1, False), (2, False), (2, True)])
@pytest.mark.parametrize("ni,nlon,nlat", [(48, 144, 90), (96, 24, 12), (48, 6, 3)])
def test_option_sweep_entry_parallel_keeps_the_bits(fg, gpu_ok, order, mono, ni, nlon, nlat):
    """k_apply_epx (weight, cell_measures, --target_grid, missing values, gradient mask, the monotone limiter's values) against the
    lane-per-row kernel k_apply_ex (fg_set_apply_ep(0)) on rows of ~4, ~300 and ~2000 exchange cells; the oracle pins the
    lane-per-row semantics in test_sweep_every_option_bitwise, whose cases run through k_apply_epx as well."""
    import torch
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    p = fg.XgridPlan.create(order, grids, fg.GridConfig(nlon, nlat, lo, la))
    a_in, a_out = p.get_cell_area(nlon * nlat)
    p.finalize()
    nc = 6 * ni * ni
    nf = 6 * (ni + 2) ** 2 if order == 2 else nc
    rng = np.random.default_rng(31)
    missing = 1.0e20
    src = rng.standard_normal(nf); src[rng.random(nf) < 0.15] = missing
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    gx, gy = t(4.0 * rng.standard_normal(nc)), t(4.0 * rng.standard_normal(nc))
    gm = t((rng.random(nc) < 0.3).astype(np.int32))
    w = t(rng.uniform(0.2, 1.0, nc)); fa = t(np.asarray(a_in) * rng.uniform(0.3, 1.0, nc)); ca = t(np.asarray(a_in)); cao = t(np.asarray(a_out))
    res = []
    try:
        for ep in (0, 1):
            fg.lib().fg_set_apply_ep(ep)
            out = torch.full((nlon * nlat,), np.nan, dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()                   # (the fill runs on torch's stream, the library on the plan's own)
            g = p.apply_ex(t(src), out, nz=1, grad_x_t=gx if order == 2 else None, grad_y_t=gy if order == 2 else None,
                           grad_mask_t=gm if order == 2 else None, has_missing=True, missing=missing, weight_t=w,
                           field_area_t=fa, area_missing=-1e20, cell_area_in_t=ca, cell_area_out_t=cao, monotonic=mono, want_gsum=True)
            p.sync()
            res.append((out.cpu().numpy(), g))
    finally:
        fg.lib().fg_set_apply_ep(1)
        p.destroy()
    assert np.array_equal(res[0][0].view(np.uint64), res[1][0].view(np.uint64))
    assert np.float64(res[0][1]).view(np.uint64) == np.float64(res[1][1]).view(np.uint64)


@pytest.mark.parametrize("ni,nlon,nlat", [(96, 90, 45), (96, 24, 12), (48, 6, 3)])
def test_first_order_eight_level_sweep_long_rows_keeps_the_bits(fg, gpu_ok, ni, nlon, nlat):
    """First order, 8 levels, fine -> coarse: k_apply_ep8g<1> (chunked entry-parallel tiles) against the row-serial k_apply_il
    (fg_set_apply_ep(0)); rows of ~20, ~300, ~2000 exchange cells."""
    import torch
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    p = fg.XgridPlan.create(1, [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)], fg.GridConfig(nlon, nlat, lo, la))
    p.finalize()
    src = torch.from_numpy(np.random.default_rng(4).standard_normal((8, 6 * ni * ni))).to("cuda:0")
    outs = []
    try:
        for ep in (0, 1):
            fg.lib().fg_set_apply_ep(ep)
            out = torch.empty((8, nlon * nlat), dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            g = p.apply(src, out, nz=8, want_gsum=True); p.sync()
            outs.append((out.cpu().numpy(), g))
    finally:
        fg.lib().fg_set_apply_ep(1)
        p.destroy()
    assert np.isfinite(outs[0][0]).all()
    assert np.array_equal(outs[0][0].view(np.uint64), outs[1][0].view(np.uint64))
    assert np.float64(outs[0][1]).view(np.uint64) == np.float64(outs[1][1]).view(np.uint64)


@pytest.mark.parametrize("ni,nlon,nlat", [(12, 360, 180), (64, 8, 4), (48, 36, 18)])
@pytest.mark.parametrize("order", [1, 2])
def test_extreme_resolution_ratios_vs_oracle(fg, gpu_ok, order, ni, nlon, nlat):
    """The paths only mismatched resolutions reach, end to end against the oracle: a coarse source over a fine target (every source
    cell listed for the wave-per-cell candidate scan, ~150 candidate pairs per cell: big-cell compaction) and a fine source over a
    very coarse target (destination rows of ~300 and ~3000 exchange cells: one atomic per run of lanes in the clip, rows ranked
    tile by tile in the CSR build, chunked entry-parallel sweeps) -- exchange cells, areas, di / dj and the remapped field of one
    level and of eight, bit for bit."""
    import torch
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    o = orc.orc_setup(order, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    p = fg.XgridPlan.create(order, grids, fg.GridConfig(nlon, nlat, lo, la))
    p.finalize()
    check_xgrid(p.get_xgrid(), o, order, True)
    for nz in (1, 8):
        data, gx, gy = make_fields(ni, lon, lat, nz, order, seed=5 + nz)
        pack = lambda arrs: torch.from_numpy(np.ascontiguousarray(np.stack(
            [np.concatenate([a[k].ravel() for a in arrs]) for k in range(nz)]))).to("cuda:0")
        out = torch.empty(nz * nlon * nlat, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        p.apply(pack(data), out, nz=nz, grad_x_t=pack(gx) if order == 2 else None, grad_y_t=pack(gy) if order == 2 else None)
        p.sync()
        ref, _ = orc.orc_apply(order, o, [ni] * 6, [ni] * 6, [d.reshape(nz, -1) for d in data],
                               [g.reshape(nz, -1) for g in gx] if order == 2 else None,
                               [g.reshape(nz, -1) for g in gy] if order == 2 else None, None, False, -1e20, nlon, nlat, nz)
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(np.asarray(ref).reshape(-1))), nz
    p.destroy()
