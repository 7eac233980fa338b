"""Search + finalize over a matrix of grid pairs (resolution ratios, curvilinear targets, great circle): ms per step and exchange cells / s,
to spot paths nobody tuned.  usage: matrix_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
dev = "cuda:0"
h2d = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)


def grid(kind):
    if kind[0] == "C":
        ni = int(kind[1:]); lon, lat = fg.gnomonic_ed_corners(ni)
        return [(ni, ni, lon[t], lat[t]) for t in range(6)]
    if kind[0] == "T":                                   # tripolar nlon x nlat
        nx, ny = (int(v) for v in kind[1:].split("x")); lon, lat = fg.tripolar_corners(nx, ny)
        return [(nx, ny, lon, lat)]
    nx, ny = (int(v) for v in kind[1:].split("x")); lon, lat = fg.latlon_corners(nx, ny)
    return [(nx, ny, lon, lat)]


def run(src, dst, order, gc=False):
    gs = grid(src); gd = grid(dst)
    if gc:
        xin = [tuple(h2d(a) for a in fg.latlon2xyz(g[2], g[3])) for g in gs]
        dsts = [(nxo, nyo, tuple(h2d(a) for a in fg.latlon2xyz(lo, la))) for (nxo, nyo, lo, la) in gd]
    else:
        lon_t = [h2d(g[2]) for g in gs]; lat_t = [h2d(g[3]) for g in gs]
        dsts = [(nxo, nyo, h2d(lo), h2d(la)) for (nxo, nyo, lo, la) in gd]
    nxs, nys = [g[0] for g in gs], [g[1] for g in gs]
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); tot = 0
        for d in dsts:
            if gc:
                p = fg.XgridPlan.create_great_circle_dev(nxs, nys, xin, d[0], d[1], d[2], np.pi / max(d[1], 1), 2 * np.pi / max(d[0], 1))
            else:
                p = fg.XgridPlan.create_dev(order, nxs, nys, lon_t, lat_t, d[0], d[1], d[2], d[3], 0.0, 0.0)
            p.finalize(); p.sync(); tot += p.nxgrid; p.destroy()
        ts.append(time.perf_counter() - t0)
    dt = min(ts[1:])
    print(f"{src:>10s} -> {dst:<10s} order {order}{' gc' if gc else '   '}: {dt * 1e3:9.3f} ms  nxgrid {tot:9d}  {tot / dt / 1e9:6.2f} e9 cells/s  ({len(dsts)} destination tile(s), inputs resident)", flush=True)


for src, dst, order, gc in [("C384", "L1440x720", 2, False), ("C384", "L1440x720", 1, False), ("T1440x1080", "C384", 1, False), ("C384", "C96", 2, False),
                            ("C96", "C384", 2, False), ("L1440x720", "C384", 1, False), ("L360x180", "L1440x720", 1, False), ("L1440x720", "L360x180", 1, False),
                            ("T360x200", "L360x180", 1, False), ("C96", "L360x180", 1, True), ("C384", "L180x90", 1, True), ("C48", "L720x360", 1, True)]:
    try:
        run(src, dst, order, gc)
    except Exception as e:
        print(f"{src} -> {dst} order {order} gc {gc}: FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
