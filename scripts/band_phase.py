"""Per-phase device times (HIP events) of the banded search for a few (ranks, rank) choices, culling on: where a rank's time goes
when its band is an eighth / a quarter / all of the target.  usage: python scripts/band_phase.py"""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_package
fg = load_package()
ni, nlon, nlat = 384, 1440, 720
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]; lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
fg.lib().fg_set_search_cull(1); fg.lib().fg_set_profiling(1)
for N, r in ((8, 0), (8, 3), (4, 1), (1, 0)):
    j0, j1 = fg.band_rows(nlat, N, r)
    blo = torch.from_numpy(np.ascontiguousarray(lo[j0:j1 + 1])).to(dev); bla = torch.from_numpy(np.ascontiguousarray(la[j0:j1 + 1])).to(dev)
    acc = {}
    for rep in range(6):
        p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, j1 - j0, blo, bla, np.pi / nlat, 2 * np.pi / nlon)
        p.finalize(); p.sync()
        if rep:
            for k, v in p.phase_ms().items(): acc[k] = acc.get(k, 0) + v / 5
        st = p.stats(); n = p.nxgrid; p.destroy()
    print(N, r, n, {k: round(v, 3) for k, v in acc.items()}, st["pairs"], st["heavy"])
