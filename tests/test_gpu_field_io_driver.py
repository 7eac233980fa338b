"""integration/field_io_hip.c (get_input_data / write_field_data on the device) + integration/conserve_interp_hip.c, executed the
way fregrid.c:1041-1075 runs the originals: oracle/_ref/field_io_driver (tests/capi/field_io_driver.c, built by oracle/Makefile
against the reference's own globals.h / fregrid_util.h / conserve_interp.h / mpp.c) writes classic-netCDF input files, remaps an
NC_FLOAT variable with conserve_order2 and a packed NC_SHORT variable (scale_factor / add_offset) with conserve_order1 level by
level and writes the output file.  Its contents must equal the Python mirror's -- halo update + grad_c2l + sweep on the device,
get_input_data's widening and write_field_data's narrowing restated in numpy -- bit for bit."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "field_io_driver")


def test_field_loop_in_c_equals_the_python_mirror(fg, gpu_ok, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/field_io_driver not built (it needs /root/reference at build time: make -C oracle)")
    import torch
    from scipy.io import netcdf_file
    ni, nlon, nlat, nz = 16, 48, 24, 3
    r = subprocess.run([EXE, str(ni), str(nlon), str(nlat), str(nz), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "field_io_driver ok" in r.stdout, r.stdout + r.stderr
    with netcdf_file(os.path.join(str(tmp_path), "out.nc"), "r", mmap=False) as f:
        got_f = f.variables["t_f"][:].copy(); got_s = f.variables["t_s"][:].copy()
    assert got_f.dtype == np.dtype(">f4") or got_f.dtype == np.float32
    assert got_f.shape == (1, nz, nlat, nlon) and got_s.shape == (1, nz, nlat, nlon)
    # --- the mirror
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    gout = fg.GridConfig(nlon, nlat, lo, la)
    p2 = fg.XgridPlan.create(2, grids, gout); p2.finalize()
    p1 = fg.XgridPlan.create(1, grids, gout); p1.finalize()
    prep = fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, fg.find_contacts([ni] * 6, [ni] * 6, lon, lat))
    j, i = np.meshgrid(np.arange(ni), np.arange(ni), indexing="ij")
    dev = "cuda:0"
    ncell = 6 * ni * ni
    for k in range(nz):
        # t_f: NC_FLOAT, conserve_order2
        src = np.concatenate([(270.0 + ((t * 7 + k * 3 + j * 5 + i * 11) % 23) * 0.375 + 0.01 * k).astype(np.float32).reshape(-1) for t in range(6)])
        s_t = torch.from_numpy(src.astype(np.float64)[None, :]).to(dev)
        halo = torch.empty(1, prep.F, dtype=torch.float64, device=dev)
        gx = torch.empty(1, ncell, dtype=torch.float64, device=dev); gy = torch.empty_like(gx)
        prep.fill_halo(s_t, halo, 1); prep.gradient(halo, 1, gx, gy); prep.sync()
        out = torch.empty(1, nlon * nlat, dtype=torch.float64, device=dev)
        p2.apply(halo, out, nz=1, grad_x_t=gx, grad_y_t=gy); p2.sync()
        want = out.cpu().numpy().reshape(nlat, nlon).astype(np.float32)
        assert np.array_equal(np.ascontiguousarray(got_f[0, k]).astype(np.float32).view(np.uint32), want.view(np.uint32)), ("t_f", k)
        # t_s: NC_SHORT with scale_factor 0.01 / add_offset 250, conserve_order1 (fregrid_util.c:2097-2123, 2376-2400)
        raw = np.concatenate([((((t * 5 + k * 7 + j * 3 + i * 13) % 4001) - 2000).astype(np.int16)).reshape(-1) for t in range(6)])
        v = raw.astype(np.float64) * 0.01 + 250.0
        out1 = torch.empty(1, nlon * nlat, dtype=torch.float64, device=dev)
        p1.apply(torch.from_numpy(v[None, :]).to(dev), out1, nz=1); p1.sync()
        w = (out1.cpu().numpy().reshape(nlat, nlon) - 250.0) / 0.01
        assert np.array_equal(np.ascontiguousarray(got_s[0, k]).astype(np.int16), np.trunc(w).astype(np.int16)), ("t_s", k)
    assert np.ptp(got_f) > 1.0 and np.ptp(got_s.astype(np.int32)) > 100            # real fields, not zeros
    p1.destroy(); p2.destroy()
