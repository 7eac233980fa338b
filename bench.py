#!/usr/bin/env python3
"""bench.py -- conservative-regrid hot path on MI355X: exchange-cells/s (+ remapped-points/s).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; fits one GPU):
  C384 cubed sphere (6 tiles, gnomonic_ed) -> 1440x720 regular lat-lon (0.25 deg), conservative_order2.

One "step" = one full weight generation for the job: exchange-grid search for all 6 source tiles
against this rank's latitude band of the target + the order-2 centroid pass + the destination-row
(CSR) build.  With N > 1 ranks the target rows are split into N bands (the reference's
fregrid_parallel decomposition, fregrid_util.c:592-597); the only exchange is that of the per-source-cell
(area, clon, clat) sums of the cells present on several ranks, handed from rank to rank in the reference's
order (RCCL broadcasts of a short list, conserve_interp.c:203-221).  Fixed total work => "strong".

value = steps * (sum over ranks of nxgrid) / max-over-ranks wall time, inputs resident in HBM.
A second timed region measures the sweep (do_scalar_conserve_interp) on nz levels: remapped-points/s.

Run: python bench.py --gpus 1 --steps 20 --warmup 3
     python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synth_fields(lont, latt, nz):
    """Deterministic smooth field on the source tiles, f_k = (1 + k/8) * (2 + sin(lon_c) cos(lat_c)) at the T-cell
    centres, interior only: [nz][6*ni*ni].  Halo and gradients are produced on the device (C2lPrep)."""
    f = 2.0 + np.sin(lont) * np.cos(latt)
    return np.ascontiguousarray(np.stack([(1.0 + 0.125 * k) * f.reshape(-1) for k in range(nz)]))


def cpu_baseline(fg, lon, lat, lo, la, ni, nlon, nlat, rows):
    """The reference algorithm (brute-force pair scan) on the host on a bounded sample: `rows` source rows of tile 1
    against the full target on ONE thread (the reference's default build); the reference's own OpenMP build on all host
    cores gets a sample sized for >= 5 s.  Uses the reference's own code compiled in place (oracle/_ref) when that library
    is present, else our bit-identical C port (oracle/)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    j0 = ni // 2 - rows // 2
    sub_lon = np.ascontiguousarray(lon[0][j0:j0 + rows + 1]); sub_lat = np.ascontiguousarray(lat[0][j0:j0 + rows + 1])
    t0 = time.time()
    if orc.ref_available():
        r = orc.ref_create_xgrid(2, ni, rows, nlon, nlat, sub_lon, sub_lat, lo, la)
        kind = "reference"
    else:
        r = orc.orc_create_xgrid(2, ni, rows, nlon, nlat, sub_lon, sub_lat, lo, la)
        kind = "port"
    dt = time.time() - t0
    omp = None
    omp_path = os.path.join(ROOT, "oracle", "_ref", "libfrenc_ref_omp.so")
    if os.path.exists(omp_path):
        # the reference's own OpenMP build of the same sources (gcc -fopenmp, oracle/Makefile), all host cores, same sample
        import ctypes as C
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count()
        os.environ.setdefault("OMP_NUM_THREADS", str(ncores))           # read by libgomp when the library is loaded
        ncores = int(os.environ["OMP_NUM_THREADS"])
        L = C.CDLL(omp_path)
        dp, ip, cip = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)
        L.create_xgrid_2dx2d_order2.argtypes = [cip] * 4 + [dp] * 5 + [ip] * 4 + [dp] * 3
        L.create_xgrid_2dx2d_order2.restype = C.c_int
        cap = 5000000
        ints = [np.empty(cap, dtype=np.int32) for _ in range(4)]
        dbl = [np.empty(cap) for _ in range(3)]
        f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64).ravel()
        # sample sized from the one-thread rate for ~6 s on all cores (at most the whole tile)
        orows = int(min(ni, max(rows, rows * 6.0 / max(dt, 1e-3) * ncores * 0.04)))
        oj0 = (ni - orows) // 2
        o_lon = np.ascontiguousarray(lon[0][oj0:oj0 + orows + 1]); o_lat = np.ascontiguousarray(lat[0][oj0:oj0 + orows + 1])
        arrs = [f64(o_lon), f64(o_lat), f64(lo), f64(la), np.ones(ni * orows)]
        t1 = time.time()
        n_omp = L.create_xgrid_2dx2d_order2(C.byref(C.c_int(ni)), C.byref(C.c_int(orows)), C.byref(C.c_int(nlon)), C.byref(C.c_int(nlat)),
                                            *[a.ctypes.data_as(dp) for a in arrs], *[a.ctypes.data_as(ip) for a in ints],
                                            *[a.ctypes.data_as(dp) for a in dbl])
        dto = time.time() - t1
        omp = {"value": n_omp / dto, "unit": "exchange-cells/s", "cores": ncores, "kind": "reference (OpenMP build)",
               "seconds": dto, "nxgrid": int(n_omp), "sample": f"rows {oj0}..{oj0 + orows - 1} of tile 1 ({orows}x{ni} source cells)"}
    return {"value": r["n"] / dt, "unit": "exchange-cells/s", "cores": 1, "kind": kind, "all_cores": omp,
            "sample": f"create_xgrid_2dx2d_order2, C{ni} tile 1 rows {j0}..{j0 + rows - 1} ({rows}x{ni} source cells) "
                      f"x full {nlon}x{nlat} target: {r['n']} exchange cells in {dt:.2f} s",
            "seconds": dt}, r, j0


def gc_roofline(clip_ms, scale=1.0):
    """The great-circle clip (k_gc_screen + k_gc_solve + k_gc_walk [+ k_gc_clip_list beside it]) against the vector issue rate: the
    solve is integer arithmetic (a software x87 for the reference's long double), the walk FP64 + the exact acosl.  Instruction
    counts from the committed PMC passes (profiles/pmc_traffic.json, C384 -> 1440x720; `scale` = this job's exchange cells over that
    launch's 4 160 160 for the C768 job), time = this run's HIP-event span of the clip."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not clip_ms or not os.path.exists(pmc):
        return None
    try:
        tr = json.load(open(pmc))
    except Exception:
        return None
    ks = [k for k in ("k_gc_screen", "k_gc_solve", "k_gc_walk") if tr.get(k + "_valu_insts")]
    if not ks:
        return None
    valu_peak = 1024 * 2.4e9
    insts = scale * sum(tr[k + "_valu_insts"] for k in ks)
    out = {"kernels": ks, "instruction_scale": scale, "dominant": "k_gc_walk", "bound": "valu", "unit": "SIMD issue cycles/s", "peak": valu_peak, "clip_ms": clip_ms,
           "achieved": 4.0 * insts / (clip_ms * 1e-3), "traffic": sum(tr.get(k, 0) for k in ks) or None,
           "per_kernel": {k: {"valu_insts": tr[k + "_valu_insts"], "fp64_share": tr.get(k + "_fp64_insts", 0) / tr[k + "_valu_insts"],
                              "int_share": tr.get(k + "_int_insts", 0) / tr[k + "_valu_insts"], "lanes_active_per_valu_inst": tr.get(k + "_lanes_active")} for k in ks},
           "source": f"profiles/pmc_traffic.json (static: rocprofv3 --pmc passes of {tr.get('round', 'an earlier round')}, not collected in this run)"}
    out["frac"] = out["achieved"] / valu_peak
    return out


def pcie_leg(fg, ni, nlon, nlat, lon, lat, lo, la, device):
    """The same job through the HOST-pointer API (what a B1 / B2 caller with host arrays pays): corner arrays up, search,
    finalize, and the exchange cells (8 arrays, indices decomposed on the device) back into host memory."""
    grids = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
    gout = fg.GridConfig(nlon, nlat, lo, la)
    best = None
    for it in range(3):
        t0 = time.perf_counter()
        p = fg.XgridPlan.create(2, grids, gout, device=device)
        p.finalize(); p.sync()
        t1 = time.perf_counter()
        xg = p.get_xgrid()
        t2 = time.perf_counter()
        n = p.nxgrid
        p.destroy()
        if best is None or t2 - t0 < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1, n, len(xg["area"]))
    return {"workload": f"C{ni} -> {nlon}x{nlat} order 2, host corner arrays in, exchange cells out to host memory",
            "ms_upload_search_finalize": best[1] * 1e3, "ms_exchange_cells_to_host": best[2] * 1e3, "ms_total": best[0] * 1e3,
            "exchange_cells_per_s": best[3] / best[0], "nxgrid": best[3],
            "note": "best of 3; pageable host arrays (the reference's malloc'ed Interp_config arrays)"}


def config4_leg(fg, torch, dev, device):
    """BASELINE config 4: C768 -> 2880x1440.  (4a) create_xgrid_great_circle semantics, first order (the reference refuses
    order 2 with great circle, fregrid.c:763-765); (4b) legacy clip, second order.  Search + finalize, inputs resident."""
    ni, nlon, nlat = 768, 2880, 1440
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    h2d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = {"workload": f"C{ni} (6 tiles) -> {nlon}x{nlat} lat-lon"}
    lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
    lo_t, la_t = h2d(lo), h2d(la)
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon, device=device)
        p.finalize(); p.sync()
        ts.append(time.perf_counter() - t0); n = p.nxgrid
        p.destroy()
    out["4b_legacy_order2"] = {"nxgrid": n, "ms_per_step": min(ts[1:]) * 1e3, "exchange_cells_per_s": n / min(ts[1:])}
    th = time.perf_counter()
    xin = [tuple(h2d(a) for a in fg.latlon2xyz(lon[t], lat[t])) for t in range(6)]
    xout = tuple(h2d(a) for a in fg.latlon2xyz(lo, la))
    t_xyz = time.perf_counter() - th
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = fg.XgridPlan.create_great_circle_dev([ni] * 6, [ni] * 6, xin, nlon, nlat, xout, np.pi / nlat, 2 * np.pi / nlon, device=device)
        p.finalize(); p.sync()
        ts.append(time.perf_counter() - t0); n = p.nxgrid
        clip_ms = p.phase_ms().get("clip_general")
        p.destroy()
    out["4a_great_circle_order1"] = {"nxgrid": n, "ms_per_step": min(ts[1:]) * 1e3, "exchange_cells_per_s": n / min(ts[1:]),
                                     "host_latlon2xyz_and_upload_ms": t_xyz * 1e3, "clip_kernel_ms": clip_ms,
                                     "roofline_gc": gc_roofline(clip_ms, n / 4160160.0)}
    del lon_t, lat_t, lo_t, la_t, xin, xout
    fg.lib().fg_pool_release()
    return out


def config5_leg(fg, torch, dev, device, nz=50, nt=20, nfields=3):
    """BASELINE config 5: tripolar ocean 1440x1080 -> C384 mosaic (6 output tiles), conservative_order1 (order 2 is illegal
    for a one-tile input mosaic, fregrid.c:695-696), weights READ from cached remap files, then `nfields` 3-D fields x `nt`
    time steps x `nz` levels streamed from page-locked host memory as NC_FLOAT through the six plans and back as NC_FLOAT
    (fg_sweep): remapped-points/s INCLUDING both PCIe directions."""
    import tempfile
    nxs, nys, no = 1440, 1080, 384
    lon_s, lat_s = fg.tripolar_corners(nxs, nys)
    lon_d, lat_d = fg.gnomonic_ed_corners(no)
    gin = [fg.GridConfig(nxs, nys, lon_s, lat_s)]
    tmp = tempfile.mkdtemp(prefix="fg_remap_")
    t0 = time.perf_counter()
    nx_tot = 0
    for t in range(6):                                           # WRITE pass: search once, cache the weights
        p = fg.XgridPlan.create(1, gin, fg.GridConfig(no, no, lon_d[t], lat_d[t]), device=device)
        x = p.get_xgrid()
        fg.write_remap_file(os.path.join(tmp, f"remap.tile{t + 1}.nc"), 1, x["t_in"], x["i_in"], x["j_in"], x["i_out"], x["j_out"], x["area"])
        nx_tot += p.nxgrid
        p.destroy()
    t_write = time.perf_counter() - t0
    t0 = time.perf_counter()
    plans = []
    for t in range(6):                                           # READ pass (conserve_interp.c:62-126)
        r = fg.read_remap_file(os.path.join(tmp, f"remap.tile{t + 1}.nc"), 1)
        p = fg.XgridPlan.create_empty(1, [nxs], [nys], no, no, device=device)
        p.set_xgrid(r["t_in"], r["i_in"], r["j_in"], r["i_out"], r["j_out"], r["area"])
        plans.append(p)
    t_read = time.perf_counter() - t0
    ncell = nxs * nys
    hin = fg.HostBuffer((nz, ncell), np.float32)
    houts = [fg.HostBuffer((nz, no * no), np.float32) for _ in range(6)]
    base = (280.0 + 15.0 * np.cos(np.linspace(0, 8 * np.pi, ncell))).astype(np.float32)
    for k in range(nz):
        hin.array[k] = base * np.float32(1.0 - 0.004 * k)
    sw = fg.Sweep(plans, None, np.float32, np.float32)
    outs = [h.array for h in houts]
    sw.run(hin.array, outs)                                       # warm-up (allocations, first touch)
    t0 = time.perf_counter()
    for _ in range(nt * nfields):
        sw.run(hin.array, outs)
    dtw = time.perf_counter() - t0
    pts = 6 * no * no * nz * nt * nfields
    bytes_in, bytes_out = ncell * nz * 4 * nt * nfields, 6 * no * no * nz * 4 * nt * nfields
    chk = float(np.mean(outs[2][0]))
    # the same sweep with everything resident (no transfers), for the ratio
    src_t = torch.from_numpy(np.ascontiguousarray(hin.array[:8].astype(np.float64))).to(dev)
    out_t = torch.empty(8, no * no, dtype=torch.float64, device=dev)
    for p in plans:
        p.apply(src_t, out_t, nz=8)
    torch.cuda.synchronize(); [p.sync() for p in plans]
    t0 = time.perf_counter()
    for _ in range(20):
        for p in plans:
            p.apply(src_t, out_t, nz=8)
    [p.sync() for p in plans]
    dtr = (time.perf_counter() - t0) / 20
    sw.destroy(); hin.free(); [h.free() for h in houts]
    for p in plans:
        p.destroy()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    return {"workload": f"tripolar {nxs}x{nys} -> C{no} (6 tiles), conservative_order1, cached remap files, {nfields} fields x {nt} steps x {nz} levels, NC_FLOAT in / out",
            "nxgrid": nx_tot, "remapped_points_per_s_incl_transfers": pts / dtw, "seconds": dtw,
            "pcie_GBps_in": bytes_in / dtw / 1e9, "pcie_GBps_out": bytes_out / dtw / 1e9,
            "remapped_points_per_s_resident": 6 * no * no * 8 / dtr,
            "weights_search_and_write_s": t_write, "weights_read_and_csr_s": t_read, "check_mean_tile3_level0": chk,
            "note": "page-locked host buffers (fg_host_alloc); levels cross the link as float and are widened / narrowed on the device"}


def banded_search_job(fg, torch, dist, dev, local_rank, world, rank, ni, nlon, nlat, steps, warmup, repeats):
    """One strong-scaling weight-generation job: C<ni> -> nlon x nlat, order 2, this rank's latitude band of the target (equal
    rows: measured max/mean <= 1.06 up to 8 ranks, scripts/band_time.py), source cells culled to the band, and the product's
    exchange: the running sums of the source cells present on several ranks handed from rank to rank (parallel.CellSumExchange,
    bit-reproducible; schedule built once).  Returns median seconds per region of `steps` steps, max over ranks."""
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    j0, j1 = fg.band_rows(nlat, world, rank)
    h2d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lon_t = [h2d(lon[t]) for t in range(6)]; lat_t = [h2d(lat[t]) for t in range(6)]
    lo_t, la_t = h2d(lo[j0:j1 + 1]), h2d(la[j0:j1 + 1])
    stream = torch.cuda.current_stream().cuda_stream
    mk = lambda: fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, j1 - j0, lo_t, la_t, np.pi / nlat, 2 * np.pi / nlon,
                                         device=local_rank, stream=stream)
    ex = None
    if world > 1:
        fg.lib().fg_set_search_cull(1)
        p0 = mk(); ex = fg.CellSumExchange([p0], str(dev)); p0.destroy()
    plan = [None]

    def step():
        if plan[0] is not None:
            plan[0].destroy()
        p = mk()
        if world > 1:
            total = ex.run([p], complete=False)
            p.finalize(total.data_ptr())
        else:
            p.finalize(None)
        plan[0] = p
        return p
    fg.lib().fg_set_profiling(0)
    for _ in range(warmup):
        step()
    reps = []
    for _ in range(repeats):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            p = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        reps.append(time.perf_counter() - t0)
    nx = torch.tensor([float(plan[0].nxgrid)], dtype=torch.float64, device=dev)
    rt = torch.tensor(reps, dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(nx); dist.all_reduce(rt, op=dist.ReduceOp.MAX)
    plan[0].destroy()
    fg.lib().fg_set_search_cull(0); fg.lib().fg_set_profiling(1)
    reps = [float(v) for v in rt.cpu()]
    med = float(np.median(reps))
    return {"workload": f"C{ni} (6 tiles) -> {nlon}x{nlat}, conservative_order2, search + centroid pass + CSR build per step",
            "nxgrid": int(nx.item()), "ms_per_step": med / steps * 1e3, "ms_per_step_min": min(reps) / steps * 1e3,
            "exchange_cells_per_s": steps * int(nx.item()) / med, "steps": steps, "repeats": repeats, "n_gpus": world,
            "shared_cells_handed_over": (ex.nsh if ex is not None else 0)}


def banded_gc_job(fg, torch, dist, dev, local_rank, world, rank, ni, nlon, nlat, steps, warmup):
    """BASELINE config 4 under ranks: C<ni> -> nlon x nlat with create_xgrid_great_circle semantics (first order), this rank's band of
    the target, source cells culled to the band by their bounding caps; search + finalize per step (no data-path collective:
    first order has no per-source-cell sums), then ONCE the WRITE branch -- exchange cells to the host, gathered on the root in rank
    order, one remap file (conserve_interp.c:368-445) -- timed separately.  Unit vectors come from the host libm as the reference's
    latlon2xyz makes them (not timed)."""
    import tempfile
    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    j0, j1 = fg.band_rows(nlat, world, rank)
    h2d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    th = time.perf_counter()
    xin = [tuple(h2d(a) for a in fg.latlon2xyz(lon[t], lat[t])) for t in range(6)]
    xout = tuple(h2d(a) for a in fg.latlon2xyz(lo[j0:j1 + 1], la[j0:j1 + 1]))
    t_xyz = time.perf_counter() - th
    stream = torch.cuda.current_stream().cuda_stream
    fg.lib().fg_set_search_cull(1 if world > 1 else 0)
    plan = [None]

    def step():
        if plan[0] is not None:
            plan[0].destroy()
        p = fg.XgridPlan.create_great_circle_dev([ni] * 6, [ni] * 6, xin, nlon, j1 - j0, xout, np.pi / nlat, 2 * np.pi / nlon,
                                                  device=local_rank, stream=stream)
        p.finalize(None)
        plan[0] = p
        return p
    fg.lib().fg_set_profiling(0)
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        p = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    fg.lib().fg_set_search_cull(0); fg.lib().fg_set_profiling(1)
    # WRITE: every rank's exchange cells to its host, gathered, one file on the root
    tmp = tempfile.mkdtemp(prefix="fg_remap_gc_") if rank == 0 else None
    if world > 1:
        box = [tmp]
        dist.broadcast_object_list(box, src=0)
        tmp = box[0]
    path = os.path.join(tmp, "remap_gc.nc")
    tw = time.perf_counter()
    x = p.get_xgrid()
    ic = fg.InterpConfig(nxgrid=p.nxgrid, i_in=x["i_in"], j_in=x["j_in"], i_out=x["i_out"], j_out=x["j_out"], t_in=x["t_in"], area=x["area"],
                         remap_file=path)
    g = fg.GridConfig(nlon, j1 - j0, None, None); g.isc, g.jsc = 0, j0
    n_glob = fg.write_remap_gathered(ic, g, 1)
    t_write = time.perf_counter() - tw
    size = os.path.getsize(path) if rank == 0 else 0
    if rank == 0:
        os.remove(path); os.rmdir(tmp)
    tt = torch.tensor([dt, t_write], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt, t_write = float(tt[0]), float(tt[1])
    p.destroy(); plan[0] = None
    del xin, xout
    fg.lib().fg_pool_release()
    return {"workload": f"C{ni} (6 tiles) -> {nlon}x{nlat}, create_xgrid_great_circle semantics (first order), search + CSR build per step, "
                        "then the gathered remap-file write once",
            "nxgrid": n_glob, "ms_per_step": dt / steps * 1e3, "exchange_cells_per_s": steps * n_glob / dt, "steps": steps, "n_gpus": world,
            "remap_write_s": t_write, "remap_file_bytes": size, "host_latlon2xyz_and_upload_ms": t_xyz * 1e3,
            "note": "remap_write_s = device -> host copy of this rank's exchange cells + gather on the root (rank order) + file write; max over ranks"}


class _DeviceDoubles:
    """__cuda_array_interface__ view of n doubles at a device pointer: torch.as_tensor aliases it (no copy)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def exchange_in_place(fg, torch, plan, bidx_t, ncell, dev):
    """The boundary cells' (area, clon, clat) sums all-reduced IN the plan's own array (no 21 MB copy, no host sync: the plan's
    stream is torch's current stream, and torch orders the collective after the search and before the centroid pass)."""
    sums = torch.as_tensor(_DeviceDoubles(plan.cell_sums_ptr(), 3 * ncell), device=dev)
    fg.allreduce_cell_sums_sparse(sums, bidx_t, ncell)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ni", type=int, default=384, help="C<ni> source cubed sphere")
    ap.add_argument("--nlon", type=int, default=1440)
    ap.add_argument("--nlat", type=int, default=720)
    ap.add_argument("--nz", type=int, default=8, help="levels per sweep launch")
    ap.add_argument("--apply-steps", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each (median reported)")
    ap.add_argument("--no-phase-timing", action="store_true", help="skip the separate pass that records per-phase HIP events")
    ap.add_argument("--legs", default="all", help="comma list of extra legs: gc,pcie,config4,config5,cpu,c768,c768gc,config5_full (or all / none); "
                                                  "all = the first five; with --gpus N > 1 the c768 jobs always run")
    ap.add_argument("--cpu-rows", type=int, default=32, help="source rows in the CPU baseline sample (0 = skip)")
    ap.add_argument("--gc-steps", type=int, default=3, help="timed great-circle searches of the same grids (N=1 only; 0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    fg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # FG_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks sharing one GPU (development only;
    # the driver's multi-GPU runs use one rank per GPU over RCCL == backend "nccl")
    backend = os.environ.get("FG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    ni, nlon, nlat, nz = args.ni, args.nlon, args.nlat, args.nz
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    j0, j1 = fg.band_rows(nlat, world, rank)
    ny_band = j1 - j0
    lon_t = [torch.from_numpy(lon[t]).to(dev) for t in range(6)]
    lat_t = [torch.from_numpy(lat[t]).to(dev) for t in range(6)]
    lo_t = torch.from_numpy(np.ascontiguousarray(lo[j0:j1 + 1])).to(dev)
    la_t = torch.from_numpy(np.ascontiguousarray(la[j0:j1 + 1])).to(dev)
    ncell_in = 6 * ni * ni
    src_h = synth_fields(lont, latt, nz)
    src_t = torch.from_numpy(src_h).to(dev)
    out_t = torch.empty(nz * ny_band * nlon, dtype=torch.float64, device=dev)
    # order-2 input preparation on the device: halo update across the cube edges + grad_c2l
    prep = fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, fg.find_contacts([ni] * 6, [ni] * 6, lon, lat), device=local_rank)
    prep.set_stream(torch.cuda.current_stream().cuda_stream)
    data_t = torch.empty(nz, prep.F, dtype=torch.float64, device=dev)
    gx_t = torch.empty(nz, ncell_in, dtype=torch.float64, device=dev)
    gy_t = torch.empty_like(gx_t)

    def prepare():
        prep.fill_halo(src_t, data_t, nz)
        prep.gradient(data_t, nz, gx_t, gy_t)

    for _ in range(3):
        prepare()
    torch.cuda.synchronize()
    tp = time.perf_counter()
    prep_steps = 20
    for _ in range(prep_steps):
        prepare()
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - tp) / prep_steps
    total_sums = torch.empty(3 * ncell_in, dtype=torch.float64, device=dev)
    mean_dlat, mean_dlon = np.pi / nlat, 2 * np.pi / nlon
    stream = torch.cuda.current_stream().cuda_stream
    legs = {"gc", "pcie", "config4", "config5", "cpu"} if args.legs == "all" else set(x for x in args.legs.split(",") if x and x != "none")

    def barrier():
        if world > 1:
            dist.barrier()

    plan = [None]
    # communication schedule of the decomposition (built once, like the band extents): the source cells with exchange cells on
    # more than one rank are the only ones whose sums need an exchange
    bidx_t = None
    a_in_full = None
    exchange_check = None
    ex = None
    mk = lambda: fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, ny_band, lo_t, la_t, mean_dlat, mean_dlon,
                                         device=local_rank, stream=stream)
    if world > 1:
        lo_full, la_full = torch.from_numpy(lo).to(dev), torch.from_numpy(la).to(dev)
        pf = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, nlat, lo_full, la_full, mean_dlat, mean_dlon,
                                     device=local_rank, stream=stream)        # the un-banded job: the single-rank sums, for the check
        cs = pf.get_cell_struct(0, ncell_in)          # 0 = source cells
        a_in_full = np.asarray(pf.get_cell_area(nlon * nlat)[0]).copy()       # (a culled plan reports area 0 for the cells it skipped)
        serial = torch.empty(3 * ncell_in, dtype=torch.float64, device=dev)
        pf.copy_cell_sums(serial)
        pf.destroy(); del lo_full, la_full
        bidx = fg.boundary_source_cells(cs["lat_min"], cs["lat_max"], la, nlat, world)
        bidx_t = torch.from_numpy(bidx.astype(np.int64)).to(dev)
        fg.lib().fg_set_search_cull(1)          # each rank builds records only for the source cells that can meet its band
        pc = mk()
        ex = fg.CellSumExchange([pc], str(dev))
        # one-off check of the exchange the timed steps use: for every source cell with exchange cells on this rank the handed-over
        # sums must carry the BITS of the single-rank search's sums (bands in rank order = ascending destination index)
        total = ex.run([pc], complete=False)
        loc = torch.empty(3 * ncell_in, dtype=torch.float64, device=dev)
        pc.copy_cell_sums(loc)
        mine3 = (loc[:ncell_in] != 0).repeat(3)
        exchange_check = {"cells_on_this_rank": int(mine3.sum().item()) // 3, "shared_cells": ex.nsh,
                          "bit_identical_to_single_rank_sums": bool(torch.equal(total[mine3], serial[mine3]))}
        torch.cuda.synchronize()
        pc.destroy(); del serial, loc, total

    def step(mode="ordered"):
        if plan[0] is not None:
            plan[0].destroy()
        p = mk()
        if world > 1 and mode == "ordered":
            total = ex.run([p], complete=False)
            p.finalize(total.data_ptr())
        else:
            if world > 1:
                exchange_in_place(fg, torch, p, bidx_t, ncell_in, dev)
            p.finalize(None)
        plan[0] = p
        return p

    # ---- headline: K steps per timed region, profiling events OFF; the region is repeated to show the spread
    fg.lib().fg_set_profiling(0)
    # one rank, one destination tile: the plan's own per-cell sums are the totals, so the search queues its finalize work itself
    # (fg_set_search_finalize; what setup_conserve_interp does in that case) -- same kernels, one host round trip less per step
    fused = world == 1
    fg.lib().fg_set_search_finalize(1 if fused else 0)
    for _ in range(args.warmup):
        step()
    reps = []
    for rep in range(max(1, args.repeats)):
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            p = step()
        torch.cuda.synchronize(); barrier()
        reps.append(time.perf_counter() - t0)
    fg.lib().fg_set_search_finalize(0)
    two_call_ms = None
    if fused:                # the same steps as two calls (fg_plan_create_dev, then fg_plan_finalize): what a multi-tile / multi-rank job does
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        two_call_ms = (time.perf_counter() - t0) / args.steps * 1e3
    alt_ms = None
    if world > 1:            # the same steps with the cheaper, NOT bit-reproducible exchange (one sparse all-reduce of partial sums)
        for _ in range(2):
            step("allreduce")
        barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            step("allreduce")
        torch.cuda.synchronize(); barrier()
        alt_t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(alt_t, op=dist.ReduceOp.MAX)
        alt_ms = float(alt_t[0]) / args.steps * 1e3
        step()
    p = plan[0]
    nx_local = p.nxgrid
    stats = p.stats()
    # ---- per-phase device times: a separate pass with HIP events on the plan's streams (they cost ~0.1 ms per step)
    phase_acc = {}
    if not args.no_phase_timing:
        fg.lib().fg_set_profiling(1)
        for it in range(6):
            p = step()
            if it:
                for k, v in p.phase_ms().items():
                    phase_acc[k] = phase_acc.get(k, 0.0) + v / 5.0
    fg.lib().fg_set_profiling(1)
    reps_t = torch.tensor(reps, dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(reps_t, op=dist.ReduceOp.MAX)
    reps = [float(v) for v in reps_t.cpu()]
    dt = float(np.median(reps))
    fg.lib().fg_set_search_cull(0)

    trace = (lambda m: print(f"[trace rank {rank}] {m}", file=sys.stderr, flush=True)) if os.environ.get("FG_BENCH_TRACE") else (lambda m: None)
    trace("search done")
    # ---- sweep leg
    apply_steps = args.apply_steps
    for _ in range(3):
        p.apply(data_t, out_t, nz=nz, grad_x_t=gx_t, grad_y_t=gy_t)
        torch.cuda.synchronize(); trace("apply warm-up")
    p.phase_ms()
    barrier(); torch.cuda.synchronize()
    ta = time.perf_counter()
    for _ in range(apply_steps):
        p.apply(data_t, out_t, nz=nz, grad_x_t=gx_t, grad_y_t=gy_t)
    torch.cuda.synchronize(); barrier()
    dta = time.perf_counter() - ta
    apply_call_ms = p.phase_ms()["apply"]            # device time of one level-major call (2 transposes + sweep)
    # one level per call: what fregrid's level loop does (fregrid.c:1045-1061: do_scalar_conserve_interp(..., nz = 1))
    out1_t = out_t[:ny_band * nlon]
    for _ in range(3):
        p.apply(data_t, out1_t, nz=1, grad_x_t=gx_t, grad_y_t=gy_t)
    barrier(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(apply_steps):
        p.apply(data_t, out1_t, nz=1, grad_x_t=gx_t, grad_y_t=gy_t)
    torch.cuda.synchronize(); barrier()
    dt1 = time.perf_counter() - t1
    # the sweep kernel alone, on fields kept interleaved [cell][nz]
    nb = 16 if nz >= 16 else (8 if nz >= 8 else (4 if nz >= 4 else 2))
    il = lambda t, n: t[:nb].reshape(nb, n).t().contiguous()
    data_il, gx_il, gy_il = il(data_t, data_t.shape[1]), il(gx_t, ncell_in), il(gy_t, ncell_in)
    out_il = torch.empty(ny_band * nlon, nb, dtype=torch.float64, device=dev)
    for _ in range(3):
        p.apply_interleaved(nb, data_il, out_il, gx_il, gy_il)
    p.phase_ms()
    barrier(); torch.cuda.synchronize()
    tb = time.perf_counter()
    for _ in range(apply_steps):
        p.apply_interleaved(nb, data_il, out_il, gx_il, gy_il)
    torch.cuda.synchronize(); barrier()
    dtb = time.perf_counter() - tb
    apply_kernel_ms = p.phase_ms()["apply"]
    # ... and on the merged records fg_plan_apply / fg_plan_apply_records sweep (the product path of the level-major API)
    rec_kernel_ms, dtr = 0.0, 0.0
    if nz <= 8:
        rec0_t = torch.empty(ncell_in, 3, fg.C2lPrep.records_nb(nz), dtype=torch.float64, device=dev)
        prep.records(src_t, nz, rec0_t)
        for _ in range(3):
            p.apply_records(nz, rec0_t, out_t)
        p.phase_ms()
        barrier(); torch.cuda.synchronize()
        tr0 = time.perf_counter()
        for _ in range(apply_steps):
            p.apply_records(nz, rec0_t, out_t)
        torch.cuda.synchronize(); barrier()
        dtr = time.perf_counter() - tr0
        rec_kernel_ms = p.phase_ms()["apply"]
    # ---- per-chunk pipeline of an order-2 variable: halo update + grad_c2l + sweep of nz levels, (a) through the reference's
    # level-major gradient arrays, (b) fused: one kernel from the unpadded levels to the sweep's records (fg_c2l_records +
    # fg_plan_apply_records) -- bit-identical outputs (tests/test_gpu_c2l.py)
    dtl = dtf = 0.0
    if nz <= 8:
        rec_t = torch.empty(ncell_in, 3, fg.C2lPrep.records_nb(nz), dtype=torch.float64, device=dev)

        def pipe_level_major():
            prepare()
            p.apply(data_t, out_t, nz=nz, grad_x_t=gx_t, grad_y_t=gy_t)

        def pipe_fused():
            prep.records(src_t, nz, rec_t)
            p.apply_records(nz, rec_t, out_t)

        times = []
        for fn in (pipe_level_major, pipe_fused):
            for _ in range(3):
                fn()
            barrier(); torch.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(apply_steps):
                fn()
            torch.cuda.synchronize(); barrier()
            times.append((time.perf_counter() - tq) / apply_steps)
        dtl, dtf = times
    gsum_out = p.apply(data_t, out_t, nz=1, grad_x_t=gx_t, grad_y_t=gy_t, want_gsum=True)
    # the same flux from this rank's exchange cells on the host: sum_x (f + gx*di + gy*dj)[src(x)] * area(x) -- what the
    # sweep must reproduce to rounding (conservation of the remap itself, independent of the geometric closure of the grids)
    xg = p.get_xgrid()
    s_idx = xg["t_in"].astype(np.int64) * ni * ni + xg["j_in"].astype(np.int64) * ni + xg["i_in"]
    f0 = src_h[0][s_idx] + gx_t[0].cpu().numpy()[s_idx] * xg["c1"] + gy_t[0].cpu().numpy()[s_idx] * xg["c2"]
    gsum_xgrid = float(np.sum(f0 * xg["area"]))

    # ---- reductions over ranks
    red = torch.tensor([dt1, dta, dtb, dtl, dtf, dtr], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(nx_local), float(gsum_out), gsum_xgrid], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt1, dta, dtb, dtl, dtf, dtr = (float(red[k]) for k in range(6))
    nx_total, gsum_out, gsum_xgrid = int(tot[0].item()), float(tot[1].item()), float(tot[2].item())

    if rank == 0:
        value = args.steps * nx_total / dt
        ndst = nlon * nlat
        remap_pts = apply_steps * ndst * nz / dta
        nsteps = max(args.steps, 1)
        phases = dict(phase_acc)
        # algorithmic bytes (destination corners counted once because all six source
        # tiles are searched in one pass): 16 B per corner read, 8 B per source cell (mask), 40 B per xcell written
        nx_rank0 = nx_local
        alg_search = 16.0 * (6 * (ni + 1) ** 2 + (nlon + 1) * (ny_band + 1)) + 8.0 * ncell_in + 40.0 * nx_rank0
        clip_ms = phases.get("clip_quad", 0.0)
        # The dominant kernel of the search is FP64-VALU bound (SURVEY.md §8d predicted it): its roofline is the vector issue
        # rate -- a wave64 FP64 instruction occupies its SIMD for 4 cycles, 1024 SIMDs at 2.4 GHz -- and the HBM view is secondary.
        valu_peak = 1024 * 2.4e9
        fp64_peak = 78.6e12           # vector FP64, MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
        roof = {"kernel": "k_clip_quad<2, true>", "bound": "valu", "achieved": None, "peak": valu_peak, "unit": "SIMD issue cycles/s",
                "frac": None, "traffic": None,
                "kernel_ms": clip_ms, "algorithmic_bytes_per_launch": alg_search,
                "hbm": {"achieved": (alg_search / 1e9) / (clip_ms / 1e3) if clip_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s"},
                "note": "achieved = 4 issue cycles x wave VALU instructions of one launch (SQ_INSTS_VALU, PMC pass) / live kernel time = how busy the "
                        "vector pipes are, NOT how much of that is useful FP64 work: see fp64 (flop/s against the 78.6 TF vector peak, FP64 share of the "
                        "instructions, lanes live per instruction); hbm.achieved = algorithmic bytes of the whole search / this kernel's time (the figure "
                        "SURVEY.md 8d asks for); the HBM-bound kernel of the path is the sweep: roofline_apply"}
        roof["hbm"]["frac"] = roof["hbm"]["achieved"] / HBM_PEAK_GBS if roof["hbm"]["achieved"] else None
        # sweep: weights streamed once per launch of nz levels + per level the source fields and the output
        alg_apply = 32.0 * nx_rank0 + nb * (24.0 * ncell_in + 8.0 * nlon * ny_band)
        sweep_ms = rec_kernel_ms if rec_kernel_ms > 0 else apply_kernel_ms
        roof_a = {"kernel": ("k_apply_ep8<256,256>" if nb == 8 else f"k_apply_il<2,{nb},4,MERGED>") if rec_kernel_ms > 0 else f"k_apply_il<2,{nb}>", "bound": "hbm",
                  "achieved": (alg_apply / 1e9) / (sweep_ms / 1e3) if sweep_ms > 0 else None,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                  "algorithmic_bytes_per_launch": alg_apply, "kernel_ms": sweep_ms, "levels_per_launch": nb,
                  "kernel_ms_separate_arrays": apply_kernel_ms}
        roof_a["frac"] = roof_a["achieved"] / HBM_PEAK_GBS if roof_a["achieved"] else None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        tr = {}
        if os.path.exists(pmc):
            try:
                tr = json.load(open(pmc))
            except Exception:
                tr = {}
        if tr.get("k_clip_quad_valu_insts") and tr.get("k_clip_quad_pairs") and clip_ms > 0:
            # a band's launch (world > 1): the per-pair counts of the single-GPU PMC pass, scaled by this rank's candidate pairs
            scale = 1.0 if world == 1 else stats["pairs"] / tr["k_clip_quad_pairs"]
            src = (f"profiles/pmc_traffic.json (static: rocprofv3 --pmc passes of {tr.get('round', 'an earlier round')} on the single-GPU launch, not collected "
                   "in this run" + ("" if world == 1 else "; scaled by this rank's candidate pairs / the profiled launch's") + ")")
            roof["achieved"] = 4.0 * tr["k_clip_quad_valu_insts"] * scale / (clip_ms * 1e-3)
            roof["frac"] = roof["achieved"] / valu_peak
            roof["valu_insts_source"] = src
            if tr.get("k_clip_quad_fp64_wave_flops"):
                lanes = tr.get("k_clip_quad_lanes_active")
                fl_all = 64.0 * tr["k_clip_quad_fp64_wave_flops"] * scale / (clip_ms * 1e-3)
                roof["fp64"] = {"flops_per_s_all_lanes": fl_all, "frac_of_78.6TF_all_lanes": fl_all / fp64_peak,
                                "flops_per_s_live_lanes": (fl_all * lanes / 64.0) if lanes else None,
                                "frac_of_78.6TF_live_lanes": (fl_all * lanes / 64.0 / fp64_peak) if lanes else None,
                                "fp64_share_of_valu_insts": tr.get("k_clip_quad_fp64_insts", 0) / tr["k_clip_quad_valu_insts"],
                                "int32_share_of_valu_insts": tr.get("k_clip_quad_int32_insts", 0) / tr["k_clip_quad_valu_insts"],
                                "lanes_active_per_valu_inst": lanes,
                                "note": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 (FMA = 2 flops) x 64 lanes / live kernel time; live lanes = "
                                        "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU of 64: the gap between pipe-busy and useful work is divergence "
                                        "(polygons of 3..8 vertices in one wave, flat / general edges) plus moves, selects and compares"}
            if world == 1:
                roof["traffic"] = tr.get("k_clip_quad"); roof["traffic_source"] = src
                roof_a["traffic"] = tr.get("k_apply"); roof_a["traffic_source"] = src
        # mass conservation (conserve_interp.c:874-907): input flux uses get_grid_area cell areas
        a_in = a_in_full if a_in_full is not None else np.asarray(p.get_cell_area(nlon * ny_band)[0])
        gsum_in = float(np.sum(src_h[0] * a_in))
        line = {
            "metric": "exchange-cells/s", "value": value, "unit": "exchange-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / nsteps * 1e3,
            "repeats": len(reps), "ms_per_step_min": min(reps) / nsteps * 1e3, "ms_per_step_all": [r / nsteps * 1e3 for r in reps],
            "timing_note": "median over `repeats` timed regions of `steps` steps each (max over ranks per region); HIP-event phase "
                           "timing is OFF in them, phase_ms comes from a separate pass",
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"C{ni} cubed sphere (6 tiles) -> {nlon}x{nlat} lat-lon, conservative_order2: "
                                   "exchange-grid search + centroid pass + CSR build per step",
                       "nxgrid": nx_total, "parallelism": f"{world} latitude band(s) of the target, one per GPU", "world_size": world,
                       "exchange": (None if world == 1 else f"the product's: running (area, clon, clat) sums of the {ex.nsh} source cells present on "
                                    f"several ranks ({100.0 * ex.nsh / ncell_in:.2f} % of {ncell_in}) handed from rank to rank by {world} broadcasts "
                                    "(parallel.CellSumExchange, conserve_interp.c:203-221's order: bit-identical to one rank)")},
            "exchange_check": exchange_check,
            "step_calls": ("one: the search queues its finalize work itself before its single synchronisation (fg_set_search_finalize: one destination "
                           "tile on one rank, the plan's own per-cell sums are the totals)" if fused else
                           "fg_plan_create_dev, exchange of the shared cells' sums, fg_plan_finalize(totals)"),
            "ms_per_step_two_calls": two_call_ms,
            "ms_per_step_allreduce_exchange": alt_ms,
            "allreduce_exchange_note": (None if world == 1 else "same steps with ONE sparse all-reduce of partial sums in place of the hand-over: "
                                        "cheaper, but the last bits of di / dj then depend on the rank count -- not the product's default"),
            "remapped_points_per_s": remap_pts, "apply_ms_per_call": dta / apply_steps * 1e3, "apply_levels": nz,
            "apply_single_level_ms": dt1 / apply_steps * 1e3,
            "remapped_points_per_s_single_level": apply_steps * ndst / dt1,
            "single_level_note": "fg_plan_apply with nz = 1, what the reference's level loop calls (fregrid.c:1045-1061); entry-parallel kernel k_apply_ep1",
            "apply_device_ms_per_call": apply_call_ms,
            "remapped_points_per_s_interleaved": apply_steps * ndst * nb / dtb,
            "remapped_points_per_s_records": (apply_steps * ndst * nz / dtr) if dtr > 0 else None,
            "mass_rel_err": abs(gsum_out - gsum_in) / abs(gsum_in),
            "mass_rel_err_note": "reference definition (conserve_interp.c:874-907): input flux uses get_grid_area cell areas, so it "
                                 "carries the geometric closure of the exchange grid itself, 9.6e-10 for these grids in the reference too "
                                 "(BASELINE.md); mass_rel_err_xgrid is the remap's own conservation over the exchange cells",
            "mass_rel_err_xgrid": abs(gsum_out - gsum_xgrid) / abs(gsum_xgrid),
            "prep_ms_per_call": dtp * 1e3, "prep_cells_per_s": ncell_in * nz / dtp,
            "prep_note": "halo update + grad_c2l for nz levels of all 6 tiles (device), feeds the order-2 sweep",
            "pipeline": None if dtf <= 0 else {
                "levels": nz, "level_major_ms_per_call": dtl * 1e3, "fused_ms_per_call": dtf * 1e3,
                "fused_remapped_points_per_s": ndst * nz / dtf,
                "note": "halo update + grad_c2l + order-2 sweep of one chunk of levels; fused = one kernel from the unpadded levels to "
                        "the sweep's [cell][field,grad_x,grad_y][level] records (no halo'd copy, no level-major gradients, no merge pass)"},
            "phase_ms": phases, "search_stats": stats,
            "roofline": roof, "roofline_apply": roof_a,
        }
        pass
    # ---- the larger job of BASELINE config 4 (C768 -> 0.125 deg, 16.7 M exchange cells), same decomposition: where strong scaling
    # is not bound by launch latency.  Collective: every rank takes part.
    c768 = c768gc = None
    if world > 1 or "c768" in legs:
        c768 = banded_search_job(fg, torch, dist if world > 1 else None, dev, local_rank, world, rank, 768, 2880, 1440, 5, 2, 3)
    if world > 1 or "c768gc" in legs:
        c768gc = banded_gc_job(fg, torch, dist if world > 1 else None, dev, local_rank, world, rank, 768, 2880, 1440, 3, 1)
    if rank == 0:
        if c768 is not None:
            line["c768_order2"] = c768
        if c768gc is not None:
            line["c768_great_circle"] = c768gc
        if world == 1 and args.gc_steps > 0 and "gc" in legs:
            # BASELINE config 4's clip method on the same grids (create_xgrid_great_circle semantics, first order): unit
            # vectors made on the host with libm as the reference does (not timed), search timed with inputs resident
            th = time.perf_counter()
            xyz_in = [fg.latlon2xyz(lon[t], lat[t]) for t in range(6)]
            xyz_out = fg.latlon2xyz(lo, la)
            t_xyz = time.perf_counter() - th
            xin = [tuple(torch.from_numpy(a).to(dev) for a in t) for t in xyz_in]
            xout = tuple(torch.from_numpy(a).to(dev) for a in xyz_out)
            gp = None
            gc_ph = {}
            for it in range(args.gc_steps + 1):
                if it == 1:
                    torch.cuda.synchronize(); tg = time.perf_counter()
                if gp is not None:
                    gp.destroy()
                gp = fg.XgridPlan.create_great_circle_dev([ni] * 6, [ni] * 6, xin, nlon, nlat, xout, mean_dlat, mean_dlon,
                                                          device=local_rank)
                gp.finalize(None)
                if it >= 1:
                    for k, v in gp.phase_ms().items():
                        gc_ph[k] = gc_ph.get(k, 0.0) + v / args.gc_steps
            gp.sync(); torch.cuda.synchronize()
            dtg = (time.perf_counter() - tg) / args.gc_steps
            line["great_circle"] = {"workload": f"C{ni} -> {nlon}x{nlat}, create_xgrid_great_circle semantics, first order",
                                    "nxgrid": gp.nxgrid, "ms_per_step": dtg * 1e3, "exchange_cells_per_s": gp.nxgrid / dtg,
                                    "clip_kernel_ms": gc_ph.get("clip_general"), "search_device_ms": gc_ph.get("search_total"),
                                    "host_latlon2xyz_ms": t_xyz * 1e3, "search_stats": gp.stats(),
                                    "roofline_gc": gc_roofline(gc_ph.get("clip_general"))}
            gp.destroy()
        if world == 1 and "pcie" in legs:
            line["pcie_inclusive"] = pcie_leg(fg, ni, nlon, nlat, lon, lat, lo, la, local_rank)
        if world == 1 and "config4" in legs:
            line["config4"] = config4_leg(fg, torch, dev, local_rank)
        if world == 1 and "config5" in legs:
            line["config5"] = config5_leg(fg, torch, dev, local_rank)
        if world == 1 and "config5_full" in legs:      # BASELINE config 5 at its stated size (50 fields x 365 steps x 50 levels): ~2.5 minutes, not in `all`
            line["config5_full"] = config5_leg(fg, torch, dev, local_rank, nz=50, nt=365, nfields=50)
        if world == 1 and args.cpu_rows > 0 and "cpu" in legs:
            cb, _, _ = cpu_baseline(fg, lon, lat, lo, la, ni, nlon, nlat, args.cpu_rows)
            line["cpu_baseline"] = cb
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
