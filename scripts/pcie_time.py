"""PCIe-inclusive rate of the search: host corner arrays in, host exchange-cell arrays out (the B1-style use of the library),
next to the resident-input rate bench.py reports."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_package
fg = load_package()
ni, nlon, nlat = 384, 1440, 720
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
gout = fg.GridConfig(nlon, nlat, lo, la)
best = None
for rep in range(6):
    t0 = time.perf_counter()
    p = fg.XgridPlan.create(2, grids, gout)
    p.finalize()
    t1 = time.perf_counter()
    x = p.get_xgrid()
    t2 = time.perf_counter()
    n = p.nxgrid
    p.destroy()
    if rep and (best is None or t2 - t0 < best[0]):
        best = (t2 - t0, t1 - t0, t2 - t1)
print(f"host arrays in -> search + finalize {1e3 * best[1]:.2f} ms, exchange cells back to the host {1e3 * best[2]:.2f} ms, "
      f"total {1e3 * best[0]:.2f} ms = {n / best[0]:.3e} exchange-cells/s PCIe-inclusive (nxgrid {n})")
