"""Per-phase device times (HIP events on the plan's stream) and wall time of search + finalize, C<ni> -> nlon x nlat.
usage: phase_time.py [ni nlon nlat order steps]     env FG_CULL=1: source-cell culling on; FG_BAND="N r": only band r of N of the target"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
a = [int(v) for v in sys.argv[1:]]
ni, nlon, nlat, order, steps = (a + [384, 1440, 720, 2, 30][len(a):])[:5]
fg.lib().fg_set_search_cull(int(os.environ.get("FG_CULL", "0")))
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
h2d = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
if os.environ.get("FG_BAND"):
    nb, rb = (int(v) for v in os.environ["FG_BAND"].split())
    j0, j1 = fg.band_rows(nlat, nb, rb, None)
    lo, la, nlat_full, nlat = lo[j0:j1 + 1], la[j0:j1 + 1], nlat, j1 - j0
    print(f"band {rb} of {nb}: rows {j0}..{j1}")
else:
    nlat_full = nlat
lo_t, la_t = h2d(lo), h2d(la)
torch.cuda.synchronize()


def step():
    p = fg.XgridPlan.create_dev(order, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat_full, 2 * np.pi / nlon)
    p.finalize(); p.sync()
    return p


fg.lib().fg_set_profiling(0 if os.environ.get('FG_NOPROF') else 1)
for _ in range(5):
    step().destroy()
acc = {}
for _ in range(steps):
    p = step()
    for k, v in p.phase_ms().items():
        acc[k] = acc.get(k, 0.0) + v / steps
    n = p.nxgrid; st = p.stats(); p.destroy()
fg.lib().fg_set_profiling(0)
ts = []
for _ in range(steps):
    t0 = time.perf_counter(); p = step(); ts.append(time.perf_counter() - t0); p.destroy()
ts = np.array(ts) * 1e3
print(f"C{ni} -> {nlon}x{nlat} order {order}: nxgrid {n} pairs {st['pairs']}  wall ms/step (events off) min {ts.min():.4f} median {np.median(ts):.4f}")
print("  phase ms: " + "  ".join(f"{k} {v:.4f}" for k, v in acc.items() if v > 0))
