"""How many legacy exchange-cell areas / line integrals are bit-identical to the CPU oracle (= the compiled reference)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_package
import orc
fg = load_package()
b = lambda a: np.ascontiguousarray(a).view(np.uint64)
for ni, nlon, nlat in ((48, 144, 90), (96, 360, 180)):
    lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
    for t in (0, 2):
        r = fg.create_xgrid_2dx2d_order2(ni, ni, nlon, nlat, lon[t], lat[t], lo, la)
        o = orc.orc_create_xgrid(2, ni, ni, nlon, nlat, lon[t], lat[t], lo, la)
        assert r[0] == o["n"]
        print(f"C{ni} tile {t + 1}: n={r[0]} area bits equal {np.mean(b(r[5]) == b(o['area'])):.6f} clon {np.mean(b(r[6]) == b(o['clon'])):.6f} "
              f"clat {np.mean(b(r[7]) == b(o['clat'])):.6f}  max rel area {np.max(np.abs(r[5] - o['area']) / o['area']):.2e}")
    ca = fg.get_grid_area(ni, ni, lon[2], lat[2]); cr = orc.orc_get_grid_area(ni, ni, lon[2], lat[2])
    print(f"  cell areas bits equal {np.mean(b(ca) == b(cr)):.6f}")
