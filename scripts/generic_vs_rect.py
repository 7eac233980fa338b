"""Full-size agreement of the generic bins path with the rectilinear-target path: C384 -> 1440x720 (and a polar cube tile as target
against itself through a culling search), every exchange cell bit for bit.  usage: python scripts/generic_vs_rect.py [ni nlon nlat]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_package
fg = load_package()
a = [int(v) for v in sys.argv[1:]]
ni, nlon, nlat = (a + [384, 1440, 720][len(a):])[:3]
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]


def run(order, gin, gout, rect=1, cull=0):
    fg.lib().fg_set_search_rect(rect); fg.lib().fg_set_search_cull(cull)
    try:
        p = fg.XgridPlan.create(order, gin, gout); p.finalize(); x = p.get_xgrid(); st = p.stats(); p.destroy()
    finally:
        fg.lib().fg_set_search_rect(1); fg.lib().fg_set_search_cull(0)
    return x, st


def same(x, y):
    return all(np.array_equal(x[k].view(np.uint8), y[k].view(np.uint8)) for k in x)


for order in (1, 2):
    xr, _ = run(order, grids, fg.GridConfig(nlon, nlat, lo, la), 1)
    xg, st = run(order, grids, fg.GridConfig(nlon, nlat, lo, la), 0)
    print(f"C{ni} -> {nlon}x{nlat} order {order}: nxgrid {len(xr['area'])}, generic == rectilinear: {same(xr, xg)} (heavy {st['heavy']})", flush=True)
    assert same(xr, xg)
src = [fg.GridConfig(nlon, nlat, lo, la)]
for t in (0, 2, 5):
    x0, st = run(2, src, fg.GridConfig(ni, ni, lon[t], lat[t]), 1, 0)
    x1, _ = run(2, src, fg.GridConfig(ni, ni, lon[t], lat[t]), 1, 1)
    per_cell = np.bincount(x0["j_out"].astype(np.int64) * ni + x0["i_out"], weights=x0["area"], minlength=ni * ni)
    p = fg.XgridPlan.create(1, src, fg.GridConfig(ni, ni, lon[t], lat[t])); a_in, a_out = p.get_cell_area(ni * ni); p.destroy()
    gap = np.max(np.abs(per_cell / a_out - 1))
    print(f"{nlon}x{nlat} -> C{ni} tile {t + 1} order 2: nxgrid {len(x0['area'])}, culling == plain: {same(x0, x1)}, "
          f"max |sum of exchange areas / cell area - 1| = {gap:.2e} (heavy {st['heavy']}, deferred {st['deferred']})", flush=True)
    assert same(x0, x1) and gap < 2e-3
