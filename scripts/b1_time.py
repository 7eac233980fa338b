"""Wall time of the B1 drop-in calls with host arrays (what a program linked against libfregrid_hip.so in place of libfrencutils sees):
create_xgrid_2dx2d_order1/2 per cubed-sphere tile, get_grid_area, create_xgrid_great_circle.  usage: b1_time.py [ni nlon nlat]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
fg = ge.load_package()
L = fg.lib()
a = [int(v) for v in sys.argv[1:]]
ni, nlon, nlat = (a + [384, 1440, 720][len(a):])[:3]
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dp = lambda v: v.ctypes.data_as(C.POINTER(C.c_double)); ip = lambda v: v.ctypes.data_as(C.POINTER(C.c_int))
ci = lambda v: C.byref(C.c_int(v))
maxx = L.get_maxxgrid()
ii, ji, io, jo = (np.empty(maxx, dtype=np.int32) for _ in range(4))
xa, xl, xt = (np.empty(maxx) for _ in range(3))
mask = np.ones(ni * ni)
for name, order in (("create_xgrid_2dx2d_order1", 1), ("create_xgrid_2dx2d_order2", 2), ("create_xgrid_great_circle", 3)):
    f = getattr(L, name); f.restype = C.c_int
    for t in (0, 2):
        lt, at = np.ascontiguousarray(lon[t]), np.ascontiguousarray(lat[t])
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            if order == 1:
                n = f(ci(ni), ci(ni), ci(nlon), ci(nlat), dp(lt), dp(at), dp(lo), dp(la), dp(mask), ip(ii), ip(ji), ip(io), ip(jo), dp(xa))
            else:
                n = f(ci(ni), ci(ni), ci(nlon), ci(nlat), dp(lt), dp(at), dp(lo), dp(la), dp(mask), ip(ii), ip(ji), ip(io), ip(jo), dp(xa), dp(xl), dp(xt))
            ts.append(time.perf_counter() - t0)
        print(f"{name} C{ni} tile {t + 1} -> {nlon}x{nlat}: {min(ts[1:]) * 1e3:.2f} ms (first call {ts[0] * 1e3:.1f}), nxgrid {n}", flush=True)
area = np.empty(nlon * nlat)
L.get_grid_area.restype = None
for rep in range(3):
    t0 = time.perf_counter(); L.get_grid_area(ci(nlon), ci(nlat), dp(lo), dp(la), dp(area)); dt = time.perf_counter() - t0
print(f"get_grid_area {nlon}x{nlat}: {dt * 1e3:.2f} ms")
