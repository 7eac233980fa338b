"""The N > 1 product path on the GPU: several ranks -- fresh child processes sharing one device, gloo for the
collectives -- run setup_conserve_interp on their latitude bands of the target and do_scalar_conserve_interp on a field; the
concatenated result must equal the single-rank one BIT FOR BIT, including the di / dj of the source cells cut by a band
boundary (the reference gathers the exchange cells and adds them in rank order "for the purpose of bitwise reproducing",
conserve_interp.c:203-221; parallel.ordered_cell_sums hands one running total from rank to rank)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NI, NLON, NLAT, NZ = 48, 144, 90, 3


def _inputs(fg):
    lon, lat = fg.gnomonic_ed_corners(NI)
    lo, la = fg.latlon_corners(NLON, NLAT)
    rng = np.random.default_rng(21)
    data = [rng.standard_normal((NZ, NI + 2, NI + 2)) + 3.0 for _ in range(6)]
    gx = [rng.standard_normal((NZ, NI, NI)) for _ in range(6)]
    gy = [rng.standard_normal((NZ, NI, NI)) for _ in range(6)]
    return lon, lat, lo, la, data, gx, gy


def _run(fg, world, rank, outdir):
    import torch
    lon, lat, lo, la, data, gx, gy = _inputs(fg)
    j0, j1 = fg.band_rows(NLAT, world, rank)
    grid_in = [fg.GridConfig(NI, NI, lon[t], lat[t]) for t in range(6)]
    g = fg.GridConfig(NLON, j1 - j0, np.ascontiguousarray(lo[j0:j1 + 1]), np.ascontiguousarray(la[j0:j1 + 1]))
    g.isc, g.jsc = 0, j0
    interp = [fg.InterpConfig()]
    fg.setup_conserve_interp(6, grid_in, 1, [g], interp, fg.CONSERVE_ORDER2)
    ic = interp[0]
    plan = ic.plan
    dev = "cuda:0"
    ncell = 6 * NI * NI
    src = torch.from_numpy(np.concatenate([d.reshape(NZ, -1) for d in data], axis=1)).to(dev)
    gxt = torch.from_numpy(np.concatenate([d.reshape(NZ, -1) for d in gx], axis=1)).to(dev)
    gyt = torch.from_numpy(np.concatenate([d.reshape(NZ, -1) for d in gy], axis=1)).to(dev)
    out = torch.empty(NZ, NLON * (j1 - j0), dtype=torch.float64, device=dev)
    plan.apply(src, out, nz=NZ, grad_x_t=gxt, grad_y_t=gyt); plan.sync()
    np.savez(os.path.join(outdir, f"r{world}_{rank}.npz"), t_in=ic.t_in, i_in=ic.i_in, j_in=ic.j_in, i_out=ic.i_out, j_out=ic.j_out + j0,
             area=ic.area, di=ic.di_in, dj=ic.dj_in, out=out.cpu().numpy())
    plan.destroy()


def _worker(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import load_package
    fg = load_package()
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        _run(fg, world, rank, outdir)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bands_on_several_ranks_equal_the_single_rank_plan_bit_for_bit(fg, gpu_ok, tmp_path, world):
    import torch.multiprocessing as mp
    td = str(tmp_path)
    _run(fg, 1, 0, td)
    mp.spawn(_worker, args=(world, os.path.join(td, "init"), td), nprocs=world, join=True)
    one = np.load(os.path.join(td, "r1_0.npz"))
    parts = [np.load(os.path.join(td, f"r{world}_{r}.npz")) for r in range(world)]
    ni2 = NI * NI
    key = lambda x: (x["t_in"].astype(np.int64) * ni2 + x["j_in"] * NI + x["i_in"]) * NLON * NLAT + x["j_out"].astype(np.int64) * NLON + x["i_out"]
    cat = {k: np.concatenate([p[k] for p in parts]) for k in ("t_in", "i_in", "j_in", "i_out", "j_out", "area", "di", "dj")}
    k1, kp = key(one), key(cat)
    assert k1.size == kp.size and np.array_equal(np.sort(k1), np.sort(kp))
    o1, op = np.argsort(k1, kind="stable"), np.argsort(kp, kind="stable")
    bits = lambda a: np.ascontiguousarray(a).view(np.uint64)
    for k in ("area", "di", "dj"):
        assert np.array_equal(bits(one[k][o1]), bits(cat[k][op])), k
    # some source cells really are shared between ranks (otherwise the test proves nothing)
    src = one["t_in"].astype(np.int64) * ni2 + one["j_in"] * NI + one["i_in"]
    band = np.searchsorted([fg.band_rows(NLAT, world, r)[1] for r in range(world)], one["j_out"], side="right")
    nb = np.bincount(src, minlength=6 * ni2) * 0
    for r in range(world):
        nb += (np.bincount(src[band == r], minlength=6 * ni2) > 0)
    if world == 3:                              # (two bands meet on the equator, which is a grid line of the cubed sphere too)
        assert (nb > 1).sum() > 50
    # the remapped field: the bands side by side are the single-rank field, bit for bit
    out = np.concatenate([p["out"].reshape(NZ, -1, NLON) for p in parts], axis=1).reshape(NZ, -1)
    assert np.array_equal(bits(out), bits(one["out"]))


# ------------------------------------------------------------------------------------------------------------------------
# great circle under ranks (BASELINE config 4's decomposition; conserve_interp.c:163-166 search per band, :368-445 gathered WRITE)
GC_NI, GC_NLON, GC_NLAT = 24, 72, 45


def _run_gc(fg, world, rank, outdir):
    lon, lat = fg.gnomonic_ed_corners(GC_NI)
    lo, la = fg.latlon_corners(GC_NLON, GC_NLAT)
    j0, j1 = fg.band_rows(GC_NLAT, world, rank)
    grid_in = [fg.GridConfig(GC_NI, GC_NI, lon[t], lat[t]) for t in range(6)]
    g = fg.GridConfig(GC_NLON, j1 - j0, np.ascontiguousarray(lo[j0:j1 + 1]), np.ascontiguousarray(la[j0:j1 + 1]))
    g.isc, g.jsc = 0, j0
    interp = [fg.InterpConfig(remap_file=os.path.join(outdir, f"remap_w{world}.nc"))]
    fg.setup_conserve_interp(6, grid_in, 1, [g], interp, fg.CONSERVE_ORDER1 | fg.GREAT_CIRCLE | fg.WRITE)
    ic = interp[0]
    src = np.concatenate([np.cos(lat[t][:-1, :-1]).reshape(-1) + 2.0 for t in range(6)])[None, :]
    fin = [fg.FieldConfig(data=src[:, t * GC_NI * GC_NI:(t + 1) * GC_NI * GC_NI].reshape(1, GC_NI, GC_NI), var=[fg.VarConfig()]) for t in range(6)]
    fout = [fg.FieldConfig()]
    fg.do_scalar_conserve_interp(interp, 0, 6, grid_in, 1, [g], fin, fout, fg.CONSERVE_ORDER1 | fg.GREAT_CIRCLE, 1)
    np.savez(os.path.join(outdir, f"gc{world}_{rank}.npz"), t_in=ic.t_in, i_in=ic.i_in, j_in=ic.j_in, i_out=ic.i_out, j_out=ic.j_out + j0,
             area=ic.area, out=fout[0].data, a_in=np.concatenate([gi.cell_area for gi in grid_in]), a_out=g.cell_area,
             culled=np.array([ic.plan.get_cell_area(GC_NLON * (j1 - j0))[0] == 0.0]).sum())
    ic.plan.destroy()


def _worker_gc(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import load_package
    fg = load_package()
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        _run_gc(fg, world, rank, outdir)
    finally:
        dist.destroy_process_group()


def test_great_circle_bands_on_two_ranks_equal_the_single_rank_plan_and_remap_file(fg, gpu_ok, tmp_path):
    """GREAT_CIRCLE | WRITE through setup_conserve_interp on 2 ranks sharing the device: exchange cells, areas, the gathered remap
    file and the remapped field equal the single-rank ones bit for bit; the ranks really culled source cells."""
    import torch.multiprocessing as mp
    td = str(tmp_path)
    world = 2
    _run_gc(fg, 1, 0, td)
    mp.spawn(_worker_gc, args=(world, os.path.join(td, "init"), td), nprocs=world, join=True)
    one = np.load(os.path.join(td, "gc1_0.npz"))
    parts = [np.load(os.path.join(td, f"gc{world}_{r}.npz")) for r in range(world)]
    ni2 = GC_NI * GC_NI
    key = lambda x: (x["t_in"].astype(np.int64) * ni2 + x["j_in"] * GC_NI + x["i_in"]) * GC_NLON * GC_NLAT + x["j_out"].astype(np.int64) * GC_NLON + x["i_out"]
    cat = {k: np.concatenate([p[k] for p in parts]) for k in ("t_in", "i_in", "j_in", "i_out", "j_out", "area")}
    k1, kp = key(one), key(cat)
    assert k1.size == kp.size > 10000 and np.array_equal(np.sort(k1), np.sort(kp))
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    assert np.array_equal(bits(one["area"][np.argsort(k1, kind="stable")]), bits(cat["area"][np.argsort(kp, kind="stable")]))
    assert int(one["culled"]) == 0 and all(int(p["culled"]) > ni2 for p in parts)          # each band leaves out more than a tile
    for p in parts:                                                                         # cell areas do not depend on the culling
        assert np.array_equal(bits(p["a_in"]), bits(one["a_in"]))
    assert np.array_equal(bits(np.concatenate([p["a_out"] for p in parts])), bits(one["a_out"]))
    out = np.concatenate([p["out"].reshape(1, -1, GC_NLON) for p in parts], axis=1)
    assert np.array_equal(bits(out), bits(one["out"].reshape(1, -1, GC_NLON)))
    # the remap files: same exchange cells (rank order in the gathered file, as the reference's mpp_gather leaves them)
    f1, f2 = fg.read_remap_file(os.path.join(td, "remap_w1.nc"), 1), fg.read_remap_file(os.path.join(td, f"remap_w{world}.nc"), 1)
    o1, o2 = np.argsort(key(f1), kind="stable"), np.argsort(key(f2), kind="stable")
    assert np.array_equal(key(f1)[o1], key(f2)[o2]) and np.array_equal(bits(f1["area"][o1]), bits(f2["area"][o2]))
    assert np.array_equal(key(f2), kp)                                                      # ... and in rank order


def test_accumulate_cell_sums_waits_for_torchs_stream(fg, gpu_ok):
    """the accumulate kernel runs on the plan's own stream; the zero fill of its running total is queued on torch's
    stream behind a long fill.  The result must be the synchronised one."""
    import torch
    lon, lat, lo, la, *_ = _inputs(fg)
    grids = [fg.GridConfig(NI, NI, lon[t], lat[t]) for t in range(6)]
    p = fg.XgridPlan.create(2, grids, fg.GridConfig(NLON, NLAT, lo, la))
    n = p.ncells_in
    want = torch.zeros(3 * n, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    p.accumulate_cell_sums(want); torch.cuda.synchronize()
    total = torch.full((3 * n,), 1.0e300, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()                   # (filled on torch's stream; the library works on the plan's own)
    big = torch.empty(1 << 28, dtype=torch.float64, device="cuda:0")          # 2 GiB: the fill takes ~1 ms
    torch.cuda.synchronize()
    big.fill_(1.0); big.mul_(2.0); total.zero_()
    p.accumulate_cell_sums(total)
    torch.cuda.synchronize()
    assert torch.equal(total, want) and float(want[:n].sum()) > 0
    p.destroy()
