import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
nlon, nlat = int(sys.argv[2]) if len(sys.argv) > 2 else 1440, int(sys.argv[3]) if len(sys.argv) > 3 else 720
t0 = time.time(); lon, lat = fg.gnomonic_ed_corners(N); lo, la = fg.latlon_corners(nlon, nlat); print("gridgen", time.time() - t0)
dev = 'cuda:0'
lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]
lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
lo_t, la_t = torch.from_numpy(lo).to(dev), torch.from_numpy(la).to(dev)
torch.cuda.synchronize()
dl = np.pi / nlat; dw = 2 * np.pi / nlon
for it in range(5):
    t0 = time.time()
    p = fg.XgridPlan.create_dev(2, [N] * 6, [N] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, dl, dw)
    t1 = time.time()
    p.finalize()
    p.sync(); t2 = time.time()
    print("search %.2f ms  finalize %.2f ms nxgrid %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, p.nxgrid), p.stats())
    if it < 4: p.destroy()
# apply
ncell = 6 * N * N
F = 6 * (N + 2) * (N + 2)
nz = 8
data = torch.randn(nz, F, dtype=torch.float64, device=dev)
gx = torch.randn(nz, ncell, dtype=torch.float64, device=dev); gy = torch.randn(nz, ncell, dtype=torch.float64, device=dev)
out = torch.empty(nz, nlat * nlon, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for it in range(5):
    t0 = time.time()
    p.apply(data, out, nz=nz, grad_x_t=gx, grad_y_t=gy); p.sync()
    print("apply nz=%d %.3f ms" % (nz, (time.time() - t0) * 1e3))
