"""fg_sweep (csrc/sweep.hip): levels streamed from host memory, in the file's type, through the plans and back must equal the
resident sweep bit for bit -- get_input_data's widening / scale / offset (fregrid_util.c:2097-2123) and write_field_data's
inverse (:2376-2406) included -- for page-locked and pageable host arrays, several chunks, order 1 and order 2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _widen(a, scale, offset, missing):
    v = a.astype(np.float64)
    if scale != 0:
        v = np.where(v != missing, v * scale, v)
    if offset != 0:
        v = np.where(v != missing, v + offset, v)
    return v


def _narrow(v, scale, offset, missing, dtype):
    v = v.copy()
    if offset != 0:
        v = np.where(v != missing, v - offset, v)
    if scale != 0:
        v = np.where(v != missing, v / scale, v)
    return v.astype(dtype)


@pytest.mark.parametrize("pinned", [True, False])
def test_streamed_order2_equals_resident(fg, gpu_ok, pinned):
    ni, nlon, nlat, nlev = 32, 96, 48, 19                        # 19 levels: chunks of 8 + 8 + 3
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    plan = fg.XgridPlan.create(2, [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)], fg.GridConfig(nlon, nlat, lo, la))
    plan.finalize()
    prep = fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, fg.find_contacts([ni] * 6, [ni] * 6, lon, lat))
    ncell = 6 * ni * ni
    rng = np.random.default_rng(7)
    src = (280.0 + 20.0 * rng.standard_normal((nlev, ncell))).astype(np.float32)
    scale, offset, missing = 0.5, 3.25, -1.0e10
    # resident reference: widen on the host exactly as get_input_data, records + sweep per chunk, narrow as write_field_data
    f64 = _widen(src, scale, offset, missing)
    ref = np.empty((nlev, nlon * nlat), dtype=np.float32)
    dev = "cuda:0"
    for l0 in range(0, nlev, 8):
        nl = min(8, nlev - l0)
        s_t = torch.from_numpy(np.ascontiguousarray(f64[l0:l0 + nl])).to(dev)
        rec = torch.empty(ncell, 3, fg.C2lPrep.records_nb(nl), dtype=torch.float64, device=dev)
        out = torch.empty(nl, nlon * nlat, dtype=torch.float64, device=dev)
        prep.records(s_t, nl, rec); prep.sync()
        plan.apply_records(nl, rec, out); plan.sync()
        ref[l0:l0 + nl] = _narrow(out.cpu().numpy(), scale, offset, missing, np.float32)
    # streamed
    sw = fg.Sweep([plan], prep, np.float32, np.float32)
    if pinned:
        hin = fg.HostBuffer((nlev, ncell), np.float32); hout = fg.HostBuffer((nlev, nlon * nlat), np.float32)
        hin.array[:] = src
        a_in, a_out = hin.array, hout.array
    else:
        a_in, a_out = src.copy(), np.empty((nlev, nlon * nlat), dtype=np.float32)
    a_out[:] = np.nan
    sw.run(a_in, [a_out], scale=scale, offset=offset, missing=missing)
    assert np.array_equal(a_out.view(np.uint32), ref.view(np.uint32))
    sw.run(a_in, [a_out], scale=scale, offset=offset, missing=missing)       # the object is reusable
    assert np.array_equal(a_out.view(np.uint32), ref.view(np.uint32))
    sw.destroy()
    if pinned:
        hin.free(); hout.free()
    plan.destroy()


def test_streamed_order1_two_output_tiles_short_input(fg, gpu_ok):
    """conserve_order1, NC_SHORT input (packed field), two output tiles sharing one upload, double output, 11 levels."""
    ni, nlev = 24, 11
    lon, lat = fg.gnomonic_ed_corners(ni)
    gin = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    outs_g = []
    for (x0, x1, y0, y1, nx, ny) in ((0.0, 180.0, -90.0, 90.0, 40, 36), (180.0, 360.0, -90.0, 90.0, 30, 20)):
        lo, la = fg.latlon_corners(nx, ny, x0, x1, y0, y1)
        outs_g.append((nx, ny, lo, la))
    plans = []
    for (nx, ny, lo, la) in outs_g:
        p = fg.XgridPlan.create(1, gin, fg.GridConfig(nx, ny, lo, la)); p.finalize(); plans.append(p)
    ncell = 6 * ni * ni
    rng = np.random.default_rng(3)
    src = rng.integers(-20000, 20000, size=(nlev, ncell)).astype(np.int16)
    scale, offset, missing = 0.01, 273.15, -32768.0
    f64 = _widen(src, scale, offset, missing)
    dev = "cuda:0"
    refs = []
    for p, (nx, ny, _, _) in zip(plans, outs_g):
        r = np.empty((nlev, nx * ny))
        for l0 in range(0, nlev, 8):
            nl = min(8, nlev - l0)
            s_t = torch.from_numpy(np.ascontiguousarray(f64[l0:l0 + nl])).to(dev)
            out = torch.empty(nl, nx * ny, dtype=torch.float64, device=dev)
            p.apply(s_t, out, nz=nl); p.sync()
            r[l0:l0 + nl] = out.cpu().numpy()
        refs.append(r)
    sw = fg.Sweep(plans, None, np.int16, np.float64)
    got = [np.empty_like(r) for r in refs]
    sw.run(src, got, scale=scale, offset=offset, missing=missing)
    for g, r in zip(got, refs):
        rr = _narrow(r, scale, offset, missing, np.float64)                  # out type double still undoes offset and scale
        assert np.array_equal(g.view(np.uint64), rr.view(np.uint64))
    sw.destroy()
    for p in plans:
        p.destroy()


def test_two_sweeps_share_plans_and_any_destruction_order(fg, gpu_ok):
    """fregrid needs one fg_sweep per pair of file types.  Two live sweeps over the same plan and gradient object,
    runs interleaved with a direct use of the plan, the plan destroyed BEFORE the sweeps: each run borrows the compute stream
    for its own duration only, results stay bit-identical and nothing touches a freed plan."""
    ni, nlon, nlat, nlev = 24, 72, 36, 10
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    plan = fg.XgridPlan.create(2, [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)], fg.GridConfig(nlon, nlat, lo, la))
    plan.finalize()
    prep = fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, fg.find_contacts([ni] * 6, [ni] * 6, lon, lat))
    ncell = 6 * ni * ni
    rng = np.random.default_rng(3)
    src32 = (280.0 + 20.0 * rng.standard_normal((nlev, ncell))).astype(np.float32)
    src64 = src32.astype(np.float64)
    sw_f = fg.Sweep([plan], prep, np.float32, np.float32)
    sw_d = fg.Sweep([plan], prep, np.float64, np.float64)
    out_f = [np.empty((nlev, nlon * nlat), dtype=np.float32) for _ in range(2)]
    out_d = [np.empty((nlev, nlon * nlat), dtype=np.float64) for _ in range(2)]
    sw_f.run(src32, [out_f[0]])
    sw_d.run(src64, [out_d[0]])
    # the plan used directly in between (its own stream again)
    dev = "cuda:0"
    rec = torch.empty(ncell, 3, fg.C2lPrep.records_nb(8), dtype=torch.float64, device=dev)
    o8 = torch.empty(8, nlon * nlat, dtype=torch.float64, device=dev)
    prep.records(torch.from_numpy(src64[:8].copy()).to(dev), 8, rec); prep.sync()
    plan.apply_records(8, rec, o8); plan.sync()
    assert np.array_equal(o8.cpu().numpy().view(np.uint64), out_d[0][:8].view(np.uint64))
    sw_d.run(src64, [out_d[1]])
    sw_f.run(src32, [out_f[1]])
    assert np.array_equal(out_d[0].view(np.uint64), out_d[1].view(np.uint64))
    assert np.array_equal(out_f[0].view(np.uint32), out_f[1].view(np.uint32))
    assert np.array_equal(out_d[0].astype(np.float32).view(np.uint32), out_f[0].view(np.uint32))     # float levels are exact in double
    plan.destroy()                                              # before its sweeps
    sw_f.destroy(); sw_d.destroy()
