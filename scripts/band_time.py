"""Per-rank time of the banded search as a function of the rank count, measured on ONE GPU by running each rank's band in
turn (no collective): what strong scaling can reach before communication, and how well the bands are balanced.
usage: python scripts/band_time.py [equal|cost] [legacy|gc] [ni=384] [N ...]
       (source-cell culling on, as bench.py --gpus N and setup_conserve_interp under ranks run it; target 3.75*ni x 1.875*ni)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_package
fg = load_package()
args = sys.argv[1:]
mode = args.pop(0) if args and args[0] in ("equal", "cost") else "equal"
clip = args.pop(0) if args and args[0] in ("legacy", "gc") else "legacy"
ni = 384
if args and args[0].startswith("ni="):
    ni = int(args.pop(0)[3:])
nlon, nlat = ni * 15 // 4, ni * 15 // 8
lon, lat = fg.gnomonic_ed_corners(ni)
lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
h2d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
if clip == "gc":
    xin = [tuple(h2d(a) for a in fg.latlon2xyz(lon[t], lat[t])) for t in range(6)]
    xo = [a.reshape(nlat + 1, nlon + 1) for a in fg.latlon2xyz(lo, la)]
else:
    lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
w = fg.row_cost(la, 90.0 / ni) if mode == "cost" else None
# second order: a rank finalizes with the TOTAL per-source-cell sums (its own + what the exchange brought, conserve_interp.c:203-221);
# the emulation takes them from one un-banded search, so that the centroid pass sees complete cells as it does under ranks
total = None
if clip != "gc":
    pf = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, h2d(lo), h2d(la), np.pi / nlat, 2 * np.pi / nlon)
    total = torch.empty(3 * 6 * ni * ni, dtype=torch.float64, device=dev)
    pf.copy_cell_sums(total); pf.destroy(); torch.cuda.synchronize()
fg.lib().fg_set_search_cull(1)
what = "create_xgrid_great_circle semantics, order 1" if clip == "gc" else "order 2"
print(f"bands: {mode}; C{ni} -> {nlon}x{nlat} {what}; wall ms of search + finalize per rank (best of 5), culling on")
for N in [int(a) for a in args] or [1, 2, 4, 8]:
    ts, ns = [], []
    for r in range(N):
        j0, j1 = fg.band_rows(nlat, N, r, w)
        if clip == "gc":
            bx = tuple(h2d(a[j0:j1 + 1]) for a in xo)
        else:
            blo, bla = h2d(lo[j0:j1 + 1]), h2d(la[j0:j1 + 1])
        torch.cuda.synchronize()
        best = None
        for rep in range(6):
            t0 = time.perf_counter()
            if clip == "gc":
                p = fg.XgridPlan.create_great_circle_dev([ni] * 6, [ni] * 6, xin, nlon, j1 - j0, bx, np.pi / nlat, 2 * np.pi / nlon)
            else:
                p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, j1 - j0, blo, bla, np.pi / nlat, 2 * np.pi / nlon)
            p.finalize(total.data_ptr() if total is not None else None); p.sync()
            wall = (time.perf_counter() - t0) * 1e3
            n = p.nxgrid
            p.destroy()
            if rep and (best is None or wall < best):
                best = wall
        ts.append(best); ns.append(n)
    ts = np.array(ts)
    print(f"N={N}: rows {[fg.band_rows(nlat, N, r, w) for r in range(N)]}")
    print(f"      ms {[round(float(t), 3) for t in ts]}  max {ts.max():.3f}  mean {ts.mean():.3f}  max/mean {ts.max() / ts.mean():.3f}  "
          f"nxgrid sum {sum(ns)}", flush=True)
