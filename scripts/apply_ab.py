"""A/B of the sweep kernel variants on one box: identity vs XCD-banded row mapping, records (MERGED) and interleaved arrays,
NB = 8 / 16.  usage: apply_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
L = fg.lib()
ni, nlon, nlat = 384, 1440, 720
lon, lat, lont, latt = fg.gnomonic_ed_grid(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
h2d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, h2d(lo), h2d(la), np.pi / nlat, 2 * np.pi / nlon)
p.finalize()
ncell = 6 * ni * ni
rng = np.random.default_rng(0)
res = {}
for nb in (8, 16):
    data = h2d(rng.standard_normal((6 * (ni + 2) ** 2, nb))); gx = h2d(rng.standard_normal((ncell, nb))); gy = h2d(rng.standard_normal((ncell, nb)))
    out = torch.empty(nlon * nlat, nb, dtype=torch.float64, device=dev)
    rec = h2d(rng.standard_normal((ncell, 3, 8))); outl = torch.empty(8, nlon * nlat, dtype=torch.float64, device=dev)
    for xcd in (64, 991, 64, 991):
        L.fg_set_apply_xcd(64); L.fg_set_apply_ep(1 if xcd > 990 else 0)
        for name, fn in (("il", lambda: p.apply_interleaved(nb, data, out, gx, gy)), ("rec", lambda: p.apply_records(8, rec, outl))):
            if name == "rec" and nb != 8: continue
            for _ in range(5): fn()
            p.sync(); t0 = time.perf_counter()
            for _ in range(100): fn()
            p.sync(); dt = (time.perf_counter() - t0) / 100
            res.setdefault((name, nb, xcd), []).append(dt * 1e3)
    ref = out.clone()
for k, v in sorted(res.items()): print(k, ["%.4f" % x for x in v])
# same bits either way
L.fg_set_apply_ep(0); L.fg_set_apply_xcd(0); p.apply_records(8, rec, outl); p.sync(); a = outl.clone()
L.fg_set_apply_ep(1); p.apply_records(8, rec, outl); p.sync(); print('ep bitwise equal:', bool(torch.equal(a, outl)))
L.fg_set_apply_xcd(1); p.apply_records(8, rec, outl); p.sync(); b1 = outl.clone()
L.fg_set_apply_xcd(64); p.apply_records(8, rec, outl); p.sync()
print("bitwise equal:", bool(torch.equal(a, outl)))
L.fg_set_apply_xcd(64); L.fg_set_apply_vec(0); L.fg_set_apply_ep(1)
