"""repro: order-2 plan of one latitude band (culling on), then the 8-level level-major sweep"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_package
fg = load_package()
ni, nlon, nlat, nz = 384, 1440, 720, 8
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]; lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
world, rank = int(sys.argv[1]), int(sys.argv[2])
ep = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fg.lib().fg_set_apply_ep(ep)
fg.lib().fg_set_search_cull(1)
j0, j1 = fg.band_rows(nlat, world, rank)
blo = torch.from_numpy(np.ascontiguousarray(lo[j0:j1 + 1])).to(dev); bla = torch.from_numpy(np.ascontiguousarray(la[j0:j1 + 1])).to(dev)
p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, j1 - j0, blo, bla, np.pi / nlat, 2 * np.pi / nlon)
p.finalize(); p.sync()
print("band", j0, j1, "nxgrid", p.nxgrid, flush=True)
rng = np.random.default_rng(0)
data = torch.from_numpy(rng.standard_normal((nz, 6 * (ni + 2) ** 2))).to(dev)
gx = torch.from_numpy(rng.standard_normal((nz, 6 * ni * ni))).to(dev); gy = torch.from_numpy(rng.standard_normal((nz, 6 * ni * ni))).to(dev)
out = torch.empty(nz, nlon * (j1 - j0), dtype=torch.float64, device=dev)
p.apply(data, out, nz=nz, grad_x_t=gx, grad_y_t=gy); p.sync()
print("apply ok", float(out.abs().max()), flush=True)
