"""Wall time of search + finalize (C384 -> 1440x720, order 2) with the package found under ROOT: same-box A/B between
builds (each build in its own process).  usage: step_time.py ROOT [steps]"""
import os, sys, time
root = os.path.abspath(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sys.path.insert(0, root)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
fg.lib().fg_set_search_cull(int(os.environ.get("FG_CULL", "0")))
ni, nlon, nlat = 384, 1440, 720
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
h2d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
lo_t, la_t = h2d(lo), h2d(la)
torch.cuda.synchronize()
def step():
    p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon)
    p.finalize(); p.sync()
    return p
for _ in range(5): step().destroy()
ts = []
for _ in range(steps):
    t0 = time.perf_counter(); p = step(); ts.append(time.perf_counter() - t0); n = p.nxgrid; p.destroy()
ts = np.array(ts) * 1e3
print(f"{root}: nxgrid {n}  ms/step min {ts.min():.4f} median {np.median(ts):.4f} mean {ts.mean():.4f}")
