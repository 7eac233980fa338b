"""Time of the level-major sweeps fg_plan_apply with nz = 1 (what fregrid's level loop calls: do_scalar_conserve_interp(..., nz = 1)), 2, 4, 8.
usage: apply1_time.py [ni nlon nlat]   (default C384 -> 1440x720)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
a = [int(v) for v in sys.argv[1:]]
ni, nlon, nlat = (a + [384, 1440, 720][len(a):])[:3]
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
h2d = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
rng = np.random.default_rng(0)
for order in (2, 1):
    p = fg.XgridPlan.create_dev(order, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, h2d(lo), h2d(la), np.pi / nlat, 2 * np.pi / nlon)
    p.finalize()
    nc = 6 * ni * ni; nf = 6 * (ni + 2) ** 2 if order == 2 else nc
    for nz in (1, 2, 4, 8):
        src = h2d(rng.standard_normal((nz, nf)))
        gx = h2d(rng.standard_normal((nz, nc))) if order == 2 else None
        gy = h2d(rng.standard_normal((nz, nc))) if order == 2 else None
        out = torch.empty(nz, nlon * nlat, dtype=torch.float64, device=dev)
        f = (lambda: p.apply(src, out, nz=nz, grad_x_t=gx, grad_y_t=gy)) if order == 2 else (lambda: p.apply(src, out, nz=nz))
        for _ in range(5): f()
        p.sync(); t0 = time.perf_counter()
        for _ in range(100): f()
        p.sync(); dt = (time.perf_counter() - t0) / 100
        W, S = (32, 24) if order == 2 else (16, 8)
        alg = p.nxgrid * W + nz * (nc * S + nlon * nlat * 8)          # CSR once per call, fields per level
        print(f"C{ni} -> {nlon}x{nlat} ({p.nxgrid / (nlon * nlat):.1f} exchange cells per row) order {order} nz {nz}: {dt * 1e3:.4f} ms per call, {dt * 1e3 / nz:.4f} per level; algorithmic {alg / 1e6:.0f} MB -> {alg / dt / 1e12:.2f} TB/s", flush=True)
    p.destroy()
