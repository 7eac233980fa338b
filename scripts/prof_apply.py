"""Profiling target: the interleaved sweep kernel alone (C384 -> 1440x720, order 2, 8 levels), a few launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
ni, nlon, nlat, nb = 384, 1440, 720, int(os.environ.get("NB", "8"))
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]; lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
lo_t, la_t = torch.from_numpy(lo).to(dev), torch.from_numpy(la).to(dev)
p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon)
p.finalize()
data = torch.randn(6 * (ni + 2) ** 2, nb, dtype=torch.float64, device=dev)
gx = torch.randn(6 * ni * ni, nb, dtype=torch.float64, device=dev); gy = torch.randn(6 * ni * ni, nb, dtype=torch.float64, device=dev)
out = torch.empty(nlon * nlat, nb, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    p.apply_interleaved(nb, data, out, gx, gy)
p.sync()
print("ok", p.nxgrid)
import time, ctypes
fg.lib().fg_set_apply_vec(int(os.environ.get("VEC", "0")))
for n in (5, 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(n): p.apply_interleaved(nb, data, out, gx, gy)
    p.sync(); dt = (time.perf_counter() - t0) / n
print("nb", nb, "ms/launch %.4f" % (dt * 1e3), "points/s %.3e" % (nlon * nlat * nb / dt))
