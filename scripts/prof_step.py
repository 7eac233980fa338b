"""Minimal profiling target (used under rocprofv3): a few search + finalize + sweep steps on C384 -> 1440x720.
usage: prof_step.py [steps] [legacy|gc]   -- inputs are made on the host (no torch kernels run under the counters)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
ni, nlon, nlat, nz = 384, 1440, 720, 8
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = sys.argv[2] if len(sys.argv) > 2 else "legacy"
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
rng = np.random.default_rng(0)
h2d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
out = h2d(np.zeros(nz * nlon * nlat))
if os.environ.get("FG_XCD"): fg.lib().fg_set_apply_xcd(int(os.environ["FG_XCD"]))
if mode == "sweep":
    # the sweep kernels alone, for the PMC passes that settle their HBM traffic: order-2 level-major call (k_merge3 +
    # k_apply_il<2,8,2,MERGED>), the same on caller-interleaved arrays (k_apply_il<2,8,2>), and an order-1 level-major call
    # whose k_interleave3<8> is a pure streaming kernel of known byte count (calibration of the counters)
    lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
    lo_t, la_t = h2d(lo), h2d(la)
    p2 = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon); p2.finalize()
    p1 = fg.XgridPlan.create_dev(1, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon); p1.finalize()
    data = h2d(rng.standard_normal((nz, 6 * (ni + 2) ** 2)))
    gx = h2d(rng.standard_normal((nz, 6 * ni * ni))); gy = h2d(rng.standard_normal((nz, 6 * ni * ni)))
    d1 = h2d(rng.standard_normal((nz, 6 * ni * ni)))
    il = lambda t: t.t().contiguous()
    data_il, gx_il, gy_il = il(data), il(gx), il(gy)
    out_il = torch.empty(nlon * nlat, nz, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for it in range(steps):
        p2.apply(data, out, nz=nz, grad_x_t=gx, grad_y_t=gy)
        p2.apply_interleaved(nz, data_il, out_il, gx_il, gy_il)
        p1.apply(d1, out, nz=nz)
    p2.sync(); p1.sync()
    print("nxgrid", p2.nxgrid, p1.nxgrid)
    sys.exit(0)
if mode == "gc":
    xin = [tuple(h2d(a) for a in fg.latlon2xyz(lon[t], lat[t])) for t in range(6)]
    xout = tuple(h2d(a) for a in fg.latlon2xyz(lo, la))
    data = h2d(rng.standard_normal((nz, 6 * ni * ni)))
else:
    lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
    lo_t, la_t = h2d(lo), h2d(la)
    data = h2d(rng.standard_normal((nz, 6 * (ni + 2) ** 2)))
    gx = h2d(rng.standard_normal((nz, 6 * ni * ni))); gy = h2d(rng.standard_normal((nz, 6 * ni * ni)))
torch.cuda.synchronize()
for it in range(steps):
    if mode == "gc":
        p = fg.XgridPlan.create_great_circle_dev([ni] * 6, [ni] * 6, xin, nlon, nlat, xout, np.pi / nlat, 2 * np.pi / nlon)
        p.finalize()
        p.apply(data, out, nz=nz)
    else:
        p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon)
        p.finalize()
        p.apply(data, out, nz=nz, grad_x_t=gx, grad_y_t=gy)
    p.sync()
    if it < steps - 1: p.destroy()
print("nxgrid", p.nxgrid)
