"""GPU parity for the order-2 input preparation: device halo fill, grad_c2l and gradient mask against the
CPU oracle (oracle/c2l_oracle.c; its gradient is pinned bit-exact to the compiled reference), then the whole
conserve_order2 flow of tests/fregrid/cubedsphere with REAL gradients: halo -> grad_c2l -> search -> sweep."""
import ctypes as C

import numpy as np
import pytest

import gridutil
import orc

pytestmark = pytest.mark.gpu
dp, ip = orc.dp, orc.ip
P = lambda a: a.ctypes.data_as(dp)
PI = lambda a: a.ctypes.data_as(ip)
KEYS = ["tile1", "tile2", "istart1", "iend1", "jstart1", "jend1", "istart2", "iend2", "jstart2", "jend2"]
GEOM = ("dx", "dy", "area", "edge_w", "edge_e", "edge_s", "edge_n", "en_n", "en_e", "vlon", "vlat")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def oracle_prepare(fg, ni, lon, lat, lont, latt, contacts, interior, nz, missing=None):
    """CPU oracle of the same preparation: halo'd data, grad_x, grad_y, grad_mask per tile."""
    L = orc.oracle()
    f = L.orc_update_halo
    f.restype = C.c_int
    f.argtypes = [C.c_int, ip, ip, C.c_int] + [ip] * 10 + [C.c_int, C.POINTER(dp)]
    g = L.orc_grad_c2l
    g.restype = None
    g.argtypes = [C.c_int, C.c_int] + [dp] * 14
    gm = L.orc_grad_mask
    gm.restype = None
    gm.argtypes = [C.c_int, C.c_int, dp, C.c_double, ip]
    nxa = np.full(6, ni, dtype=np.int32)
    cs = [np.ascontiguousarray(contacts[k]) for k in KEYS]

    def halo(arrs, nlev):
        tiles = [np.zeros((nlev, ni + 2, ni + 2)) for _ in range(6)]
        for t in range(6):
            tiles[t][:, 1:-1, 1:-1] = arrs[t].reshape(nlev, ni, ni)
        assert f(6, PI(nxa), PI(nxa), len(cs[0]), *[PI(c) for c in cs], nlev, (dp * 6)(*[P(a) for a in tiles])) == 0
        return tiles
    xt = halo([lont[t] for t in range(6)], 1)
    yt = halo([latt[t] for t in range(6)], 1)
    data = halo(interior, nz)
    gx = [np.empty((nz, ni * ni)) for _ in range(6)]
    gy = [np.empty((nz, ni * ni)) for _ in range(6)]
    mask = [np.zeros((nz, ni * ni), dtype=np.int32) for _ in range(6)]
    for t in range(6):
        info = fg.c2l_grid_info(ni, ni, xt[t][0], yt[t][0], lon[t], lat[t])
        for k in range(nz):
            lev = np.ascontiguousarray(data[t][k])
            g(ni, ni, P(lev), *[P(info[q]) for q in GEOM], P(gx[t][k]), P(gy[t][k]))
            if missing is not None:
                gm(ni, ni, P(lev), missing, PI(mask[t][k]))
    return data, gx, gy, mask, xt, yt


def test_halo_gradient_mask_bitwise(fg, gpu_ok):
    import torch
    ni, nz = 24, 3
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    contacts = fg.find_contacts([ni] * 6, [ni] * 6, lon, lat)
    prep = fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, contacts)
    rng = np.random.default_rng(2)
    interior = [rng.standard_normal((nz, ni, ni)) + 4.0 for _ in range(6)]
    missing = 1.0e20
    for t in range(6):
        interior[t][0][(np.add.outer(np.arange(ni), np.arange(ni)) % 11) == 0] = missing
    data_o, gx_o, gy_o, mask_o, xt_o, yt_o = oracle_prepare(fg, ni, lon, lat, lont, latt, contacts, interior, nz, missing)
    dev = "cuda:0"
    src = torch.from_numpy(np.ascontiguousarray(np.stack([np.concatenate([interior[t][k].ravel() for t in range(6)]) for k in range(nz)]))).to(dev)
    halo = torch.empty(nz, prep.F, dtype=torch.float64, device=dev)
    gx = torch.empty(nz, prep.ncells, dtype=torch.float64, device=dev)
    gy = torch.empty_like(gx)
    gm = torch.empty(nz, prep.ncells, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    prep.fill_halo(src, halo, nz)
    prep.gradient(halo, nz, gx, gy, gm, has_missing=True, missing=missing)
    prep.sync()
    h = halo.cpu().numpy()
    ref_h = np.stack([np.concatenate([data_o[t][k].ravel() for t in range(6)]) for k in range(nz)])
    assert np.array_equal(_bits(h), _bits(ref_h))
    cx, cy = prep.centres()
    assert np.array_equal(_bits(cx), _bits(np.concatenate([xt_o[t].ravel() for t in range(6)])))
    assert np.array_equal(_bits(cy), _bits(np.concatenate([yt_o[t].ravel() for t in range(6)])))
    for name, got, ref in (("grad_x", gx, gx_o), ("grad_y", gy, gy_o)):
        r = np.stack([np.concatenate([ref[t][k] for t in range(6)]) for k in range(nz)])
        got = got.cpu().numpy()
        ok = np.isfinite(r) & (np.abs(r) < 1e10)                 # cells touching the 1e20 missing marker overflow alike
        assert np.array_equal(_bits(got[ok]), _bits(r[ok])), name
    rm = np.stack([np.concatenate([mask_o[t][k] for t in range(6)]) for k in range(nz)])
    assert np.array_equal(gm.cpu().numpy(), rm) and rm[0].sum() > 0 and rm[1].sum() == 0
    prep.destroy()


def test_order2_flow_with_real_gradients(fg, gpu_ok):
    """C48 -> 144x90 conserve_order2, field 10*sin(lon+lat) at the cell centres (tests/create_daily_tile_files.c:144):
    device halo + grad_c2l + search + sweep against the all-CPU oracle chain; remapped field within 1e-6 relative
    (north_star bar), in practice ~1e-12; flux conserved to the reference's own closure."""
    import torch
    ni, nlon, nlat = 48, 144, 90
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    contacts = fg.find_contacts([ni] * 6, [ni] * 6, lon, lat)
    interior = [(10.0 * np.sin(lont[t] + latt[t]))[None] for t in range(6)]
    data_o, gx_o, gy_o, _, _, _ = oracle_prepare(fg, ni, lon, lat, lont, latt, contacts, interior, 1)
    o = orc.orc_setup(2, [(ni, ni, lon[t], lat[t]) for t in range(6)], [(nlon, nlat, lo, la)])
    ref, gs_ref = orc.orc_apply(2, o, [ni] * 6, [ni] * 6, [d.reshape(1, -1) for d in data_o], gx_o, gy_o, None, False, 0.0, nlon, nlat, 1)
    dev = "cuda:0"
    prep = fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, contacts)
    plan = fg.XgridPlan.create(2, [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)], fg.GridConfig(nlon, nlat, lo, la))
    plan.finalize()
    src = torch.from_numpy(np.concatenate([interior[t].ravel() for t in range(6)])[None].copy()).to(dev)
    halo = torch.empty(1, prep.F, dtype=torch.float64, device=dev)
    gx = torch.empty(1, prep.ncells, dtype=torch.float64, device=dev)
    gy = torch.empty_like(gx)
    out = torch.empty(nlon * nlat, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    prep.fill_halo(src, halo, 1)
    prep.gradient(halo, 1, gx, gy)
    prep.sync()
    gs = plan.apply(halo, out, nz=1, grad_x_t=gx, grad_y_t=gy, want_gsum=True)
    plan.sync()
    got = out.cpu().numpy()
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))
    assert abs(gs - gs_ref) < 1e-10 * np.sum(np.abs(o["area"])) * 10.0 * 1e-3
    # second order beats first order against the analytic field on the target cell centres
    lc, tc = gridutil.cell_centres(lo, la)
    exact = 10.0 * np.sin(lc + tc).ravel()
    plan1 = fg.XgridPlan.create(1, [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)], fg.GridConfig(nlon, nlat, lo, la))
    plan1.finalize()
    out1 = torch.empty(nlon * nlat, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    plan1.apply(src, out1, nz=1)
    plan1.sync()
    band = np.abs(tc.ravel()) < 1.2                       # away from the pole rows, where lon+lat varies wildly inside a cell
    e2 = np.sqrt(np.mean((got - exact)[band] ** 2)); e1 = np.sqrt(np.mean((out1.cpu().numpy() - exact)[band] ** 2))
    assert e2 < 0.6 * e1
    prep.destroy(); plan.destroy(); plan1.destroy()


@pytest.mark.parametrize("nz", [1, 2, 3, 5, 8])
def test_gradient_records_and_sweep_bitwise(fg, gpu_ok, nz):
    """fg_c2l_gradient_records + fg_plan_apply_records (halo'd levels -> merged records -> remapped levels) against the
    level-major route fg_c2l_gradient + fg_plan_apply: records hold the same field and gradient bits, the remapped levels
    and the flux sum are bit-identical."""
    import torch
    ni, nlon, nlat = 24, 72, 36
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    contacts = fg.find_contacts([ni] * 6, [ni] * 6, lon, lat)
    rng = np.random.default_rng(5 + nz)
    dev = "cuda:0"
    prep = fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, contacts)
    plan = fg.XgridPlan.create(2, [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)], fg.GridConfig(nlon, nlat, lo, la))
    plan.finalize()
    src = torch.from_numpy(rng.standard_normal((nz, prep.ncells))).to(dev)
    halo = torch.empty(nz, prep.F, dtype=torch.float64, device=dev)
    gx = torch.empty(nz, prep.ncells, dtype=torch.float64, device=dev)
    gy = torch.empty_like(gx)
    nb = fg.C2lPrep.records_nb(nz)
    rec = torch.full((prep.ncells, 3, nb), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()                   # (the fill runs on torch's stream, the library on the plan's own)
    out_a = torch.empty(nz, nlon * nlat, dtype=torch.float64, device=dev)
    out_b = torch.empty_like(out_a)
    torch.cuda.synchronize()
    prep.fill_halo(src, halo, nz)
    prep.gradient(halo, nz, gx, gy)
    prep.gradient_records(halo, nz, rec)
    prep.sync()
    r = rec.cpu().numpy()
    assert np.array_equal(_bits(r[:, 0, :nz].T), _bits(src.cpu().numpy()))
    assert np.array_equal(_bits(r[:, 1, :nz].T), _bits(gx.cpu().numpy()))
    assert np.array_equal(_bits(r[:, 2, :nz].T), _bits(gy.cpu().numpy()))
    assert np.all(r[:, :, nz:] == 0.0)
    rec1 = torch.full_like(rec, float("nan"))            # the one-pass variant straight from the unpadded levels
    torch.cuda.synchronize()                   # (the fill runs on torch's stream, the library on the plan's own)
    prep.records(src, nz, rec1)
    prep.sync()
    assert np.array_equal(_bits(rec1.cpu().numpy()), _bits(r))
    ga = plan.apply(halo, out_a, nz=nz, grad_x_t=gx, grad_y_t=gy, want_gsum=True)
    gb = plan.apply_records(nz, rec, out_b, want_gsum=True)
    plan.sync()
    assert np.array_equal(_bits(out_a.cpu().numpy()), _bits(out_b.cpu().numpy()))
    if nz == 1:                                # fg_plan_apply sums one level's rows in a differently shaped tree
        assert abs(ga - gb) <= 1e-12 * max(abs(ga), 1.0)
    else:
        assert ga == gb
    prep.destroy(); plan.destroy()
