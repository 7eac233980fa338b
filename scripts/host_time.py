"""Where does the wall time of one bench step go?  Times create_dev / finalize / destroy separately (host clocks) next to the
device phase totals.  usage: python scripts/host_time.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_package
fg = load_package()
ni, nlon, nlat = 384, 1440, 720
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lon, lat = fg.gnomonic_ed_corners(ni)
lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]; lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
lo_t, la_t = torch.from_numpy(lo).to(dev), torch.from_numpy(la).to(dev)
for prof in (1, 0):
    fg.lib().fg_set_profiling(prof)
    acc = np.zeros(4); ph = {}
    for rep in range(reps + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon)
        t1 = time.perf_counter()
        p.finalize()
        t2 = time.perf_counter()
        p.sync()
        t3 = time.perf_counter()
        if rep >= 3:
            for k, v in p.phase_ms().items():
                ph[k] = ph.get(k, 0) + v / reps
        p.destroy()
        t4 = time.perf_counter()
        if rep >= 3:
            acc += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3]) * 1e3 / reps
    print(f"profiling={prof}: create_dev {acc[0]:.3f} ms, finalize {acc[1]:.3f}, sync {acc[2]:.3f}, destroy {acc[3]:.3f}, total {acc.sum():.3f}; "
          f"device search_total {ph.get('search_total', 0):.3f} finalize {ph.get('finalize', 0):.3f}")
