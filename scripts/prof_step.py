"""Minimal profiling target: a few search + finalize + sweep steps on C384 -> 1440x720 (used under rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
ni, nlon, nlat, nz = 384, 1440, 720, 8
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]; lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
lo_t, la_t = torch.from_numpy(lo).to(dev), torch.from_numpy(la).to(dev)
data = torch.randn(nz, 6 * (ni + 2) ** 2, dtype=torch.float64, device=dev)
gx = torch.randn(nz, 6 * ni * ni, dtype=torch.float64, device=dev); gy = torch.randn(nz, 6 * ni * ni, dtype=torch.float64, device=dev)
out = torch.empty(nz * nlon * nlat, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for it in range(steps):
    p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon)
    p.finalize()
    p.apply(data, out, nz=nz, grad_x_t=gx, grad_y_t=gy)
    p.sync()
    if it < steps - 1: p.destroy()
print("nxgrid", p.nxgrid)
