"""Per-rank time of the banded search as a function of the rank count, measured on ONE GPU by running each rank's band in
turn (no collective): what strong scaling can reach before communication, and how well the bands are balanced.
usage: python scripts/band_time.py [equal|cost] [N ...]      (source-cell culling on, as bench.py --gpus N runs it)"""
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
args = sys.argv[1:]
mode = args.pop(0) if args and args[0] in ("equal", "cost") else "cost"
w = fg.row_cost(la, 90.0 / ni) if mode == "cost" else None
fg.lib().fg_set_search_cull(1)
print(f"bands: {mode}; C{ni} -> {nlon}x{nlat} order 2; wall ms of search + finalize per rank (best of 5), culling on")
for N in [int(a) for a in args] or [1, 2, 4, 8]:
    ts, ns = [], []
    for r in range(N):
        j0, j1 = fg.band_rows(nlat, N, r, w)
        blo = torch.from_numpy(np.ascontiguousarray(lo[j0:j1 + 1])).to(dev); bla = torch.from_numpy(np.ascontiguousarray(la[j0:j1 + 1])).to(dev)
        torch.cuda.synchronize()
        best = None
        for rep in range(6):
            t0 = time.perf_counter()
            p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, j1 - j0, blo, bla, np.pi / nlat, 2 * np.pi / nlon)
            p.finalize(); p.sync()
            wall = (time.perf_counter() - t0) * 1e3
            n = p.nxgrid
            p.destroy()
            if rep and (best is None or wall < best):
                best = wall
        ts.append(best); ns.append(n)
    ts = np.array(ts)
    print(f"N={N}: rows {[fg.band_rows(nlat, N, r, w) for r in range(N)]}")
    print(f"      ms {[round(float(t), 3) for t in ts]}  max {ts.max():.3f}  mean {ts.mean():.3f}  max/mean {ts.max() / ts.mean():.3f}  nxgrid sum {sum(ns)}", flush=True)
