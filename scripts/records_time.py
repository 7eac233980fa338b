"""Time of the 8-level order-2 sweep on merged records (fg_plan_apply_records), C<ni> -> nlon x nlat.  usage: records_time.py [ni nlon nlat]"""
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
p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, h2d(lo), h2d(la), np.pi / nlat, 2 * np.pi / nlon)
p.finalize()
rng = np.random.default_rng(0)
rec = h2d(rng.standard_normal((6 * ni * ni, 3, 8))); out = torch.empty(8, nlon * nlat, dtype=torch.float64, device=dev)
for rnd in range(3):
    for _ in range(5): p.apply_records(8, rec, out)
    p.sync(); t0 = time.perf_counter()
    for _ in range(200): p.apply_records(8, rec, out)
    p.sync(); dt = (time.perf_counter() - t0) / 200
    print(f"C{ni} -> {nlon}x{nlat}: {dt * 1e3:.4f} ms per 8 levels on records (nxgrid {p.nxgrid}; checksum {float(out.sum()):.9e})", flush=True)
