"""fg_set_search_finalize(1): a search that queues its own finalize work (centroid pass from the plan's own per-cell sums, CSR
records) before its one synchronisation -- conserve_interp.c:203-358 for one destination tile on one rank -- against the two-step
sequence fg_plan_create + fg_plan_finalize: same exchange cells, same di / dj, same remapped fields, bit for bit; with attempts that
have to be repeated (a target that is not rectilinear after all, capacities sized by counting)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(fg, order, grids, gout, fused, exact=False, masks=None):
    import torch
    L = fg.lib()
    L.fg_set_search_finalize(1 if fused else 0); L.fg_set_search_mode(1 if exact else 0)
    try:
        p = fg.XgridPlan.create(order, grids, gout, masks=masks)
    finally:
        L.fg_set_search_finalize(0); L.fg_set_search_mode(0)
    p.finalize(None)                                   # fused: returns at once
    st = p.stats()
    x = p.get_xgrid()                                  # after finalize: c1 / c2 are di / dj
    ni = sum(g.nx * g.ny for g in grids)
    nf = sum((g.nx + 2) * (g.ny + 2) for g in grids) if order == 2 else ni
    rng = np.random.default_rng(3)
    src = torch.from_numpy(rng.standard_normal((8, nf))).to("cuda:0")
    gx = torch.from_numpy(rng.standard_normal((8, ni))).to("cuda:0") if order == 2 else None
    gy = torch.from_numpy(rng.standard_normal((8, ni))).to("cuda:0") if order == 2 else None
    out = torch.full((8, gout.nx * gout.ny), np.nan, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()                   # (the fill runs on torch's stream, the library on the plan's own)
    if order == 2:
        p.apply(src, out, nz=8, grad_x_t=gx, grad_y_t=gy)
    else:
        p.apply(src, out, nz=8)
    p.sync()
    res = dict(x, out=out.cpu().numpy(), n=p.nxgrid, exact_mode=st["exact_mode"], bins=st["bins"])
    p.destroy()
    return res


def same(a, b, order):
    assert a["n"] == b["n"] > 0
    for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("area", "out") + (("c1", "c2") if order == 2 else ()):
        assert np.array_equal(np.ascontiguousarray(a[k]).view(np.uint64), np.ascontiguousarray(b[k]).view(np.uint64)), k


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("target", ["latlon", "cubed tile", "regional"])
def test_fused_finalize_keeps_the_bits(fg, gpu_ok, order, target):
    lon, lat = fg.gnomonic_ed_corners(48)
    grids = [fg.GridConfig(48, 48, lon[t], lat[t]) for t in range(6)]
    if target == "latlon":
        lo, la = fg.latlon_corners(144, 90); gout = fg.GridConfig(144, 90, lo, la)
    elif target == "cubed tile":                       # not rectilinear: the attempt on the rectilinear path is repeated on the generic one
        l2, a2 = fg.gnomonic_ed_corners(32); gout = fg.GridConfig(32, 32, l2[2], a2[2])
        grids = grids[:]                               # (source: all six C48 tiles)
    else:
        lo, la = fg.latlon_corners(80, 50, 230.0, 310.0, 15.0, 65.0); gout = fg.GridConfig(80, 50, lo, la)
    two = run(fg, order, grids, gout, False)
    one = run(fg, order, grids, gout, True)
    if target == "cubed tile":
        assert one["bins"] > 0
    same(one, two, order)
    cnt = run(fg, order, grids, gout, True, exact=True)          # capacities by counting: at least one repeated attempt
    assert cnt["exact_mode"] == 1
    same(cnt, two, order)


def test_fused_plan_refuses_totals(fg, gpu_ok):
    import torch
    lon, lat = fg.gnomonic_ed_corners(24)
    grids = [fg.GridConfig(24, 24, lon[t], lat[t]) for t in range(6)]
    lo, la = fg.latlon_corners(72, 36)
    fg.lib().fg_set_search_finalize(1)
    try:
        p = fg.XgridPlan.create(2, grids, fg.GridConfig(72, 36, lo, la))
    finally:
        fg.lib().fg_set_search_finalize(0)
    tot = torch.zeros(3 * 6 * 24 * 24, dtype=torch.float64, device="cuda:0")
    with pytest.raises(Exception, match="finalized by its search"):
        p.finalize(tot.data_ptr())
    p.finalize(None)
    p.destroy()


def test_setup_conserve_interp_single_tile_uses_it(fg, gpu_ok):
    """The mirror of setup_conserve_interp takes the fused path for one destination tile on one rank; results as the oracle
    tests require them elsewhere -- here: identical to the two-step plan."""
    lon, lat = fg.gnomonic_ed_corners(32)
    grids = [fg.GridConfig(32, 32, lon[t], lat[t]) for t in range(6)]
    lo, la = fg.latlon_corners(96, 48)
    gout = [fg.GridConfig(96, 48, lo, la)]
    interp = [fg.InterpConfig()]
    fg.setup_conserve_interp(6, grids, 1, gout, interp, fg.CONSERVE_ORDER2)
    ref = run(fg, 2, grids, gout[0], False)
    assert interp[0].nxgrid == ref["n"]
    assert np.array_equal(interp[0].di_in.view(np.uint64), ref["c1"].view(np.uint64))
    assert np.array_equal(interp[0].dj_in.view(np.uint64), ref["c2"].view(np.uint64))
    assert np.array_equal(interp[0].area.view(np.uint64), ref["area"].view(np.uint64))
    interp[0].plan.destroy()
