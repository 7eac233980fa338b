"""The monotone sweep (fg_plan_apply_ex with --monotonic semantics) 20 times, for rocprofv3 --stats.  usage: mono_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
ni, nlon, nlat = 384, 1440, 720
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
h2d = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
p = fg.XgridPlan.create(2, [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)], fg.GridConfig(nlon, nlat, lo, la))
a_in, a_out = p.get_cell_area(nlon * nlat); p.finalize()
nc = 6 * ni * ni; nf = 6 * (ni + 2) ** 2
rng = np.random.default_rng(0)
src = h2d(rng.standard_normal((1, nf))); gx = h2d(4 * rng.standard_normal((1, nc))); gy = h2d(4 * rng.standard_normal((1, nc)))
gm = h2d(np.zeros(nc, dtype=np.int32)); ca = h2d(np.asarray(a_in))
out = torch.empty(nlon * nlat, dtype=torch.float64, device=dev); torch.cuda.synchronize()
for _ in range(20):
    p.apply_ex(src, out, nz=1, grad_x_t=gx, grad_y_t=gy, grad_mask_t=gm, has_missing=False, cell_area_in_t=ca, monotonic=True)
p.sync()
