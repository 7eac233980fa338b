"""Time the great-circle search at a given resolution and check size-independent properties.
usage: python scripts/gc_time.py <C-N> <nlon> <nlat> [oracle_rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_package
fg = load_package()
ni, nlon, nlat = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
orows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
lon, lat = fg.gnomonic_ed_corners(ni)
lo, la = fg.latlon_corners(nlon, nlat)
t0 = time.time()
xyz_in = [fg.latlon2xyz(lon[t], lat[t]) for t in range(6)]
xyz_out = fg.latlon2xyz(lo, la)
t1 = time.time()
print(f"host latlon2xyz: {t1 - t0:.3f} s for {6 * (ni + 1) ** 2 + (nlon + 1) * (nlat + 1)} points", flush=True)
dev = "cuda:0"
xin = [tuple(torch.from_numpy(a).to(dev) for a in t) for t in xyz_in]
xout = tuple(torch.from_numpy(a).to(dev) for a in xyz_out)
torch.cuda.synchronize()
fg.lib().fg_set_profiling(1)
for rep in range(3):
    t2 = time.time()
    plan = fg.XgridPlan.create_great_circle_dev([ni] * 6, [ni] * 6, xin, nlon, nlat, xout, np.pi / nlat, 2 * np.pi / nlon)
    plan.finalize()
    plan.sync()
    t3 = time.time()
    print(f"rep {rep}: search+finalize {1e3 * (t3 - t2):.2f} ms, nxgrid {plan.nxgrid}, exact_mode {plan.stats()['exact_mode']}, phases {plan.phase_ms()}", flush=True)
    if rep < 2:
        plan.destroy()
st = plan.stats()
print("stats", st)
x = plan.get_xgrid()
R = 6371000.0
print("sum(area)/4piR^2 - 1 =", x["area"].sum() / (4 * np.pi * R * R) - 1)
a_in, a_out = plan.get_cell_area(nlon * nlat)
got = np.bincount(x["j_out"].astype(np.int64) * nlon + x["i_out"], weights=x["area"], minlength=nlon * nlat)
print("max |sum xarea per dst cell / dst cell area - 1| =", np.max(np.abs(got / a_out - 1)))
if orows:
    import orc
    for t in (0, 2):
        j0 = ni // 2 if t == 0 else 0
        o = orc.orc_create_xgrid_gc(ni, ni, nlon, nlat, lon[t], lat[t], lo, la, j1_beg=j0, j1_end=j0 + orows, capacity=4000000)
        sel = (x["t_in"] == t) & (x["j_in"] >= j0) & (x["j_in"] < j0 + orows)
        ok = (sel.sum() == o["n"] and np.array_equal(x["i_in"][sel], o["i_in"]) and np.array_equal(x["j_in"][sel], o["j_in"])
              and np.array_equal(x["i_out"][sel], o["i_out"]) and np.array_equal(x["j_out"][sel], o["j_out"]))
        rel = np.abs(x["area"][sel] - o["area"]) / o["area"] if ok else np.array([np.nan])
        print(f"tile {t + 1} rows [{j0},{j0 + orows}): oracle n {o['n']}, lists equal {ok}, max rel area {rel.max():.3e}, "
              f"bit-identical {np.mean(x['area'][sel].view(np.uint64) == o['area'].view(np.uint64)):.4f}")
