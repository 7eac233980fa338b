"""Wall times of the calls around the hot path that no benchmark leg covers (C384 -> 1440x720, order 2): flux sums, the option
sweep, the monotone sweep, a plan from exchange-cell lists (remap-file READ), exchange-cell download, polygons, the gradient
preparation object, remap file write / read.  usage: misc_time.py"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
ni, nlon, nlat = 384, 1440, 720
lon, lat, lont, latt = fg.gnomonic_ed_grid(ni); lo, la = fg.latlon_corners(nlon, nlat)
dev = "cuda:0"
h2d = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
gout = fg.GridConfig(nlon, nlat, lo, la)


def tm(label, f, n=5, sync=None):
    f(); ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f()
        if sync: sync()
        ts.append(time.perf_counter() - t0)
    print(f"{label:64s} {min(ts) * 1e3:10.3f} ms", flush=True)


t0 = time.perf_counter(); p = fg.XgridPlan.create(2, grids, gout); a_in, a_out = p.get_cell_area(nlon * nlat); p.finalize(); p.sync()
print(f"{'fg_plan_create (host arrays) + finalize, first call':64s} {(time.perf_counter() - t0) * 1e3:10.3f} ms")
nc = 6 * ni * ni; nf = 6 * (ni + 2) ** 2
rng = np.random.default_rng(0)
src = h2d(rng.standard_normal((1, nf))); gx = h2d(rng.standard_normal((1, nc))); gy = h2d(rng.standard_normal((1, nc)))
out = torch.empty(nlon * nlat, dtype=torch.float64, device=dev); torch.cuda.synchronize()
tm("fg_plan_apply, 1 level", lambda: p.apply(src, out, nz=1, grad_x_t=gx, grad_y_t=gy), sync=p.sync)
tm("fg_plan_apply, 1 level + flux sum (gsum)", lambda: p.apply(src, out, nz=1, grad_x_t=gx, grad_y_t=gy, want_gsum=True))
w = h2d(rng.uniform(0.2, 1, nc)); ca = h2d(np.asarray(a_in)); cao = h2d(np.asarray(a_out)); fa = h2d(np.asarray(a_in) * rng.uniform(0.3, 1, nc))
gm = h2d((rng.random(nc) < 0.3).astype(np.int32))
tm("fg_plan_apply_ex: weight + cell_measures + target_grid + missing", lambda: p.apply_ex(src, out, nz=1, grad_x_t=gx, grad_y_t=gy, grad_mask_t=gm, has_missing=True, missing=1e20,
                                                                                          weight_t=w, field_area_t=fa, cell_area_in_t=ca, cell_area_out_t=cao), sync=p.sync)
tm("fg_plan_apply_ex: monotone limiter", lambda: p.apply_ex(src, out, nz=1, grad_x_t=gx, grad_y_t=gy, grad_mask_t=gm, has_missing=False, cell_area_in_t=ca, monotonic=True), sync=p.sync)
x = None
def getx():
    global x
    x = p.get_xgrid()
tm("fg_plan_get_xgrid (4.16 M exchange cells to host arrays)", getx, n=3)
tm("fg_plan_get_polygons (maxv 8)", lambda: p.get_polygons(8), n=2)
def from_lists():
    q = fg.XgridPlan.create_empty(2, [ni] * 6, [ni] * 6, nlon, nlat)
    q.set_xgrid(x["t_in"], x["i_in"], x["j_in"], x["i_out"], x["j_out"], x["area"], x["c1"], x["c2"]); q.sync(); q.destroy()
tm("plan from exchange-cell lists (fg_plan_set_xgrid, READ branch)", from_lists, n=3)
from fre_nctools_amd.remap_file import write_remap_file, read_remap_file
d = tempfile.mkdtemp(); path = os.path.join(d, "remap.nc")
tm("remap file write (order 2, 4.16 M)", lambda: write_remap_file(path, 2, x["t_in"], x["i_in"], x["j_in"], x["i_out"], x["j_out"], x["area"], x["c1"], x["c2"]), n=2)
tm("remap file read", lambda: read_remap_file(path, 2), n=2)
print(f"remap file size {os.path.getsize(path) / 1e6:.0f} MB")
tm("C2lPrep (fg_c2l_create: grid info on the host, contacts, upload)", lambda: fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, fg.find_contacts([ni] * 6, [ni] * 6, lon, lat), device=0).destroy(), n=2)
