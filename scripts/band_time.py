"""Per-rank device time of the banded search as a function of the rank count, measured on ONE GPU by running each
rank's band in turn (no collective): shows what strong scaling can reach before communication.
usage: python scripts/band_time.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_package
fg = load_package()
ni, nlon, nlat = 384, 1440, 720
lon, lat = fg.gnomonic_ed_corners(ni)
lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]; lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
fg.lib().fg_set_profiling(1)
for N in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    worst = None
    tot = 0
    for r in range(N):
        j0, j1 = fg.band_rows(nlat, N, r)
        blo = torch.from_numpy(np.ascontiguousarray(lo[j0:j1 + 1])).to(dev); bla = torch.from_numpy(np.ascontiguousarray(la[j0:j1 + 1])).to(dev)
        torch.cuda.synchronize()
        best = None
        for rep in range(4):
            t0 = time.perf_counter()
            p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, j1 - j0, blo, bla, np.pi / nlat, 2 * np.pi / nlon)
            p.finalize(); p.sync()
            wall = (time.perf_counter() - t0) * 1e3
            ph = p.phase_ms(); n = p.nxgrid
            p.destroy()
            if best is None or wall < best[0]:
                best = (wall, ph, n)
        tot += best[2]
        if worst is None or best[0] > worst[0]:
            worst = (best[0], best[1], r, best[2])
    ph = worst[1]
    print(f"N={N}: slowest rank {worst[2]} wall {worst[0]:.3f} ms nxgrid {worst[3]} (sum {tot}); " +
          ", ".join(f"{k} {v:.3f}" for k, v in ph.items() if v > 0), flush=True)
