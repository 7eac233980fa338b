#!/usr/bin/env python3
"""bench.py -- conservative-regrid hot path on MI355X: exchange-cells/s (+ remapped-points/s).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; fits one GPU):
  C384 cubed sphere (6 tiles, gnomonic_ed) -> 1440x720 regular lat-lon (0.25 deg), conservative_order2.

One "step" = one full weight generation for the job: exchange-grid search for all 6 source tiles
against this rank's latitude band of the target + the order-2 centroid pass + the destination-row
(CSR) build.  With N > 1 ranks the target rows are split into N bands (the reference's
fregrid_parallel decomposition, fregrid_util.c:592-597); the only collective is the all-reduce of the
per-source-cell (area, clon, clat) sums (RCCL), conserve_interp.c:203-221.  Fixed total work => "strong".

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
    """The reference algorithm (brute-force pair scan) on the host, single thread, on a bounded sample:
    `rows` source rows of tile 1 against the full target.  Uses the reference's own code compiled in
    place (oracle/_ref) when that library is present, else our bit-identical C port (oracle/)."""
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
        arrs = [f64(sub_lon), f64(sub_lat), f64(lo), f64(la), np.ones(ni * rows)]
        t1 = time.time()
        n_omp = L.create_xgrid_2dx2d_order2(C.byref(C.c_int(ni)), C.byref(C.c_int(rows)), C.byref(C.c_int(nlon)), C.byref(C.c_int(nlat)),
                                            *[a.ctypes.data_as(dp) for a in arrs], *[a.ctypes.data_as(ip) for a in ints],
                                            *[a.ctypes.data_as(dp) for a in dbl])
        dto = time.time() - t1
        omp = {"value": n_omp / dto, "unit": "exchange-cells/s", "cores": ncores, "kind": "reference (OpenMP build)",
               "seconds": dto, "nxgrid": int(n_omp)}
    return {"value": r["n"] / dt, "unit": "exchange-cells/s", "cores": 1, "kind": kind, "all_cores": omp,
            "sample": f"create_xgrid_2dx2d_order2, C{ni} tile 1 rows {j0}..{j0 + rows - 1} ({rows}x{ni} source cells) "
                      f"x full {nlon}x{nlat} target: {r['n']} exchange cells in {dt:.2f} s",
            "seconds": dt}, r, j0


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
    ap.add_argument("--no-phase-timing", action="store_true", help="do not record per-phase HIP events in the timed region")
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
    fg.lib().fg_set_profiling(0 if args.no_phase_timing else 1)

    def barrier():
        if world > 1:
            dist.barrier()

    plan = [None]
    # communication schedule of the decomposition (built once, like the band extents): the source cells cut by a band
    # boundary are the only ones whose partial sums live on more than one rank
    bidx_t = None
    if world > 1:
        p0 = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, ny_band, lo_t, la_t, mean_dlat, mean_dlon,
                                     device=local_rank, stream=stream)
        cs = p0.get_cell_struct(0, ncell_in)          # 0 = source cells
        p0.destroy()
        bidx = fg.boundary_source_cells(cs["lat_min"], cs["lat_max"], la, nlat, world)
        bidx_t = torch.from_numpy(bidx.astype(np.int64)).to(dev)

    def step():
        if plan[0] is not None:
            plan[0].destroy()
        p = fg.XgridPlan.create_dev(2, [ni] * 6, [ni] * 6, lon_t, lat_t, nlon, ny_band, lo_t, la_t,
                                    mean_dlat, mean_dlon, device=local_rank, stream=stream)
        if world > 1:
            p.copy_cell_sums(total_sums)
            fg.allreduce_cell_sums_sparse(total_sums, bidx_t, ncell_in)
            torch.cuda.current_stream().synchronize()
            p.finalize(total_sums.data_ptr())
        else:
            p.finalize(None)
        plan[0] = p
        return p

    for _ in range(args.warmup):
        step()
    barrier(); torch.cuda.synchronize()
    phase_acc = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        p = step()
        for k, v in p.phase_ms().items():
            phase_acc[k] = phase_acc.get(k, 0.0) + v
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    p = plan[0]
    nx_local = p.nxgrid
    stats = p.stats()

    # ---- sweep leg
    apply_steps = args.apply_steps
    for _ in range(3):
        p.apply(data_t, out_t, nz=nz, grad_x_t=gx_t, grad_y_t=gy_t)
    p.phase_ms()
    barrier(); torch.cuda.synchronize()
    ta = time.perf_counter()
    for _ in range(apply_steps):
        p.apply(data_t, out_t, nz=nz, grad_x_t=gx_t, grad_y_t=gy_t)
    torch.cuda.synchronize(); barrier()
    dta = time.perf_counter() - ta
    apply_call_ms = p.phase_ms()["apply"]            # device time of one level-major call (2 transposes + sweep)
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
    red = torch.tensor([dt, dta, dtb, dtl, dtf, dtr], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(nx_local), float(gsum_out), gsum_xgrid], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt, dta, dtb, dtl, dtf, dtr = (float(red[k]) for k in range(6))
    nx_total, gsum_out, gsum_xgrid = int(tot[0].item()), float(tot[1].item()), float(tot[2].item())

    if rank == 0:
        value = args.steps * nx_total / dt
        ndst = nlon * nlat
        remap_pts = apply_steps * ndst * nz / dta
        nsteps = max(args.steps, 1)
        phases = {k: v / nsteps for k, v in phase_acc.items()}
        # algorithmic bytes (destination corners counted once because all six source
        # tiles are searched in one pass): 16 B per corner read, 8 B per source cell (mask), 40 B per xcell written
        nx_rank0 = nx_local
        alg_search = 16.0 * (6 * (ni + 1) ** 2 + (nlon + 1) * (ny_band + 1)) + 8.0 * ncell_in + 40.0 * nx_rank0
        clip_ms = phases.get("clip_quad", 0.0)
        roof = {"kernel": "k_clip_quad<2>", "bound": "hbm",
                "achieved": (alg_search / 1e9) / (clip_ms / 1e3) if clip_ms > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "algorithmic_bytes_per_launch": alg_search, "kernel_ms": clip_ms,
                "note": "FP64-VALU bound polygon clipping, not HBM bound (SURVEY.md §8d): 2.03e8 wave VALU instructions x 4 issue "
                        "cycles = 77 % of the SIMD cycles of the launch (profiles/r01_summary.md); "
                        "the HBM-bound kernel of the path is the sweep, see roofline_apply"}
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS if roof["achieved"] else None
        # sweep: weights streamed once per launch of nz levels + per level the source fields and the output
        alg_apply = 32.0 * nx_rank0 + nb * (24.0 * ncell_in + 8.0 * nlon * ny_band)
        sweep_ms = rec_kernel_ms if rec_kernel_ms > 0 else apply_kernel_ms
        roof_a = {"kernel": f"k_apply_il<2,{nb},2,MERGED>" if rec_kernel_ms > 0 else f"k_apply_il<2,{nb}>", "bound": "hbm",
                  "achieved": (alg_apply / 1e9) / (sweep_ms / 1e3) if sweep_ms > 0 else None,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                  "algorithmic_bytes_per_launch": alg_apply, "kernel_ms": sweep_ms, "levels_per_launch": nb,
                  "kernel_ms_separate_arrays": apply_kernel_ms}
        roof_a["frac"] = roof_a["achieved"] / HBM_PEAK_GBS if roof_a["achieved"] else None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and world == 1:      # the PMC passes were taken on the single-GPU launch sizes
            try:
                tr = json.load(open(pmc))
                roof["traffic"] = tr.get("k_clip_quad")
                roof_a["traffic"] = tr.get("k_apply")
                # the clip kernel's own bound: wave VALU instructions (PMC) x 4 issue cycles on SIMD16, over the SIMD
                # cycles of the live launch duration (1024 SIMDs at 2.4 GHz)
                if tr.get("k_clip_quad_valu_insts") and clip_ms > 0:
                    roof["valu_issue_frac"] = 4.0 * tr["k_clip_quad_valu_insts"] / (clip_ms * 1e-3 * 2.4e9 * 1024)
            except Exception:
                pass
        # mass conservation (conserve_interp.c:874-907): input flux uses get_grid_area cell areas
        a_in = np.concatenate([np.asarray(fg_area) for fg_area in [p.get_cell_area(nlon * ny_band)[0]]])
        gsum_in = float(np.sum(src_h[0] * a_in))
        line = {
            "metric": "exchange-cells/s", "value": value, "unit": "exchange-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / nsteps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"C{ni} cubed sphere (6 tiles) -> {nlon}x{nlat} lat-lon, conservative_order2: "
                                   "exchange-grid search + centroid pass + CSR build per step",
                       "nxgrid": nx_total, "parallelism": f"{world} latitude band(s) of the target, one per GPU",
                       "exchange": (None if world == 1 else f"all-reduce of the (area, clon, clat) sums of the {int(bidx_t.numel())} source "
                                    f"cells cut by band boundaries ({100.0 * int(bidx_t.numel()) / ncell_in:.1f} % of {ncell_in})")},
            "remapped_points_per_s": remap_pts, "apply_ms_per_call": dta / apply_steps * 1e3, "apply_levels": nz,
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
        if world == 1 and args.gc_steps > 0:
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
                                    "host_latlon2xyz_ms": t_xyz * 1e3, "search_stats": gp.stats()}
            gp.destroy()
        if world == 1 and args.cpu_rows > 0:
            cb, _, _ = cpu_baseline(fg, lon, lat, lo, la, ni, nlon, nlat, args.cpu_rows)
            line["cpu_baseline"] = cb
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
