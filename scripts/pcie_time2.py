"""get_xgrid into FRESH pageable arrays (first touch: page faults inside the copy) against arrays that were touched before."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_package
fg = load_package()
ni, nlon, nlat = 384, 1440, 720
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
p = fg.XgridPlan.create(2, grids, fg.GridConfig(nlon, nlat, lo, la)); p.finalize()
n = p.nxgrid
L = fg.lib()
ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int)); dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
for rep in range(3):
    ints = [np.empty(n, dtype=np.int32) for _ in range(5)]; dbl = [np.empty(n) for _ in range(3)]
    t0 = time.perf_counter(); L.fg_plan_get_xgrid(p._h, *[ip(a) for a in ints], *[dp(a) for a in dbl]); t1 = time.perf_counter()
    L.fg_plan_get_xgrid(p._h, *[ip(a) for a in ints], *[dp(a) for a in dbl]); t2 = time.perf_counter()
    print(f"fresh arrays {1e3 * (t1 - t0):.2f} ms, touched arrays {1e3 * (t2 - t1):.2f} ms ({166.4e-3 / (t2 - t1):.1f} GB/s)")
