"""One search + finalize + 8-level sweep at four times the size of BASELINE config 4b: C1536 -> 5760x2880 (0.0625 deg), order 2:
counts, closure, device memory.  usage: big_probe.py [ni nlon nlat]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
a = [int(v) for v in sys.argv[1:]]
ni, nlon, nlat = (a + [1536, 5760, 2880][len(a):])[:3]
t0 = time.time(); lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat); print(f"grids on the host: {time.time() - t0:.1f} s", flush=True)
dev = "cuda:0"
h2d = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]; lo_t, la_t = h2d(lo), h2d(la)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon)
    p.finalize(); p.sync(); dt = time.perf_counter() - t0
    print(f"rep {rep}: search + finalize {dt * 1e3:.2f} ms, nxgrid {p.nxgrid}, {p.nxgrid / dt / 1e9:.2f} e9 cells/s, stats {p.stats()}", flush=True)
    if rep < 2: p.destroy()
free, tot = torch.cuda.mem_get_info()
print(f"device memory in use {(tot - free) / 2**30:.1f} GiB of {tot / 2**30:.0f}")
R = 6371000.0
a_in, a_out = p.get_cell_area(nlon * nlat)
x_area = p.get_xgrid()["area"]
print("sum(xgrid area) / 4 pi R^2 - 1 =", x_area.sum() / (4 * np.pi * R * R) - 1, " sum(cell_area_in) / 4 pi R^2 - 1 =", a_in.sum() / (4 * np.pi * R * R) - 1)
nc = 6 * ni * ni
rec = torch.ones((nc, 3, 8), dtype=torch.float64, device=dev); rec[:, 1:, :] = 0.0
out = torch.empty((8, nlon * nlat), dtype=torch.float64, device=dev); torch.cuda.synchronize()
for _ in range(3): p.apply_records(8, rec, out)
p.sync(); t0 = time.perf_counter()
for _ in range(10): p.apply_records(8, rec, out)
p.sync(); dt = (time.perf_counter() - t0) / 10
print(f"8-level sweep on records: {dt * 1e3:.3f} ms; constant field preserved: max |out - 1| = {float((out - 1).abs().max()):.3e}")
