"""Multi-GPU decomposition of the regrid job (host logic, no device code).

The path shards by destination latitude band, one band per rank, exactly like the reference's
fregrid_parallel (layout 1 x npes: tools/fregrid/fregrid_util.c:592-597; extents from
mpp_compute_extent, tools/libfrencutils/mpp_domain.c:101-158).  The only data-path collective is the sum of
the per-source-cell (area, clon, clat) partial sums over ranks (conserve_interp.c:203-221); torch.distributed
provides it (backend "nccl" == RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""


def band_rows(nlat, nranks, rank, weights=None):
    """Rows [j0, j1) of the destination grid owned by `rank`.

    weights=None: nearly equal contiguous bands (mpp_compute_extent, the reference's fregrid_parallel layout).
    weights=[nlat] per-row cost estimates (row_cost): contiguous bands of nearly equal COST -- the split that minimises the
    largest band among the greedy prefix cuts; every band keeps at least one row."""
    if weights is None:
        base, extra = divmod(nlat, nranks)
        sizes = [base + (1 if k < extra else 0) for k in range(nranks)]
        j0 = sum(sizes[:rank])
        return j0, j0 + sizes[rank]
    import numpy as np
    w = np.asarray(weights, dtype=np.float64)
    assert w.shape == (nlat,) and nranks <= nlat
    c = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for k in range(1, nranks):
        j = int(np.searchsorted(c, c[-1] * k / nranks))              # first prefix reaching k/nranks of the cost
        if j > 0 and abs(c[j - 1] - c[-1] * k / nranks) < abs(c[j] - c[-1] * k / nranks):
            j -= 1
        j = min(max(j, cuts[-1] + 1), nlat - (nranks - k))            # at least one row per band
        cuts.append(j)
    cuts.append(nlat)
    return cuts[rank], cuts[rank + 1]


def row_cost(lat_out, src_cell_deg, pole_rows=8, pole_penalty=0.0):
    """Cost estimate per destination row for band_rows(weights=...): proportional to the exchange cells the row produces --
    (1 + h/hs) (1 + w cos(lat)/hs) per destination cell of height h and width w against source cells of size hs -- plus an
    optional penalty on the few rows next to a pole, whose cells meet the source cells that hold hundreds of exchange cells
    each (wave-per-cell candidate scans, big-cell compaction, pole-fixed polygons).
    lat_out: destination corner latitudes [nlat+1, nlon+1] (radians); src_cell_deg: typical source cell size in degrees
    (C<n>: 90/n).
    Measured (scripts/band_time.py, profiles/r02_band_time.txt, C384 -> 0.25 deg, source-cell culling on): EQUAL rows already
    give max/mean = 1.01 / 1.02 / 1.06 at 2 / 4 / 8 ranks -- what an equatorial band has more of in exchange cells a polar band
    has in long-running cells -- while these weights give 1.01 / 1.09 / 1.11.  bench.py therefore keeps the reference's equal
    split; the weighted split is for grids where the balance differs (regional targets, stretched grids)."""
    import numpy as np
    la = np.asarray(lat_out)
    nlat, nlon = la.shape[0] - 1, la.shape[1] - 1
    latc = 0.5 * (la[:-1, 0] + la[1:, 0])
    h = np.abs(np.diff(la[:, 0]))
    wdt = 2 * np.pi / nlon
    hs = np.deg2rad(src_cell_deg)
    w = nlon * (1.0 + h / hs) * (1.0 + wdt * np.cos(latc) / hs)
    mean = float(np.mean(w))
    for k in range(min(pole_rows, nlat)):
        if abs(la[0, 0]) > 1.5:
            w[k] += pole_penalty * mean * (pole_rows - k) / pole_rows
        if abs(la[-1, 0]) > 1.5:
            w[nlat - 1 - k] += pole_penalty * mean * (pole_rows - k) / pole_rows
    return w


def allreduce_cell_sums(total):
    """In-place sum over ranks of the [3*ncells_in] tensor; no-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(total)
    return total


def allreduce_scalar_sum(value, device="cpu"):
    """mpp_sum_double of one scalar (the global conservation sum, conserve_interp.c:902)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        return float(t.item())
    return float(value)


def world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_minmax(fmin, fmax):
    """mpp_min_double / mpp_max_double over the per-source-cell extremes of the monotone limiter
    (conserve_interp.c:672-677); in place."""
    import torch.distributed as dist
    if world_size() > 1:
        dist.all_reduce(fmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(fmax, op=dist.ReduceOp.MAX)
    return fmin, fmax


def boundary_source_cells(lat_min, lat_max, lat_out, nlat, nranks, weights=None):
    """Indices of the source cells whose latitude range meets a boundary between two destination bands.

    Only these cells can have exchange cells on more than one rank, so only their (area, clon, clat) partial sums need
    the exchange of conserve_interp.c:203-221; every other cell's sums are complete on the one rank whose band holds it
    (and zero elsewhere).  `lat_min/lat_max` [ncells_in] are the per-cell latitude ranges (fg_plan_get_cell_struct);
    `lat_out` [nlat+1, nlon+1] the destination corner latitudes.  A boundary is the corner row between two bands; its
    latitude interval is the min/max over that row (a single value for a lat-lon grid), widened by 1e-9 rad.  The list
    depends only on the grids and the decomposition and is identical on every rank."""
    import numpy as np
    flag = np.zeros(lat_min.shape, dtype=bool)
    for r in range(nranks - 1):
        j1 = band_rows(nlat, nranks, r, weights)[1]
        row = np.asarray(lat_out)[j1]
        bmin, bmax = float(row.min()) - 1e-9, float(row.max()) + 1e-9
        flag |= (lat_min <= bmax) & (lat_max >= bmin)
    return np.flatnonzero(flag)


def allreduce_cell_sums_sparse(total, idx_t, ncells):
    """Sum over ranks of the [3*ncells] tensor restricted to the cells `idx_t` (device int64 tensor from
    boundary_source_cells); the other entries keep their local values.  Equivalent to allreduce_cell_sums when every cell
    outside idx has contributions from at most one rank, at a fraction of the message size (2 % of the cells for C384 on 8
    bands)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return total
    v = total.view(3, ncells)
    part = v[:, idx_t].contiguous()
    dist.all_reduce(part)
    v[:, idx_t] = part
    return total


def _all_reduce(t):
    """all-reduce (sum) of a device tensor; through host memory when the backend is gloo (CPU rehearsals of the N > 1 path)"""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == "gloo":
        c = t.cpu()
        dist.all_reduce(c)
        t.copy_(c)
    else:
        dist.all_reduce(t)
    return t


def _broadcast(t, src):
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == "gloo":
        c = t.cpu()
        dist.broadcast(c, src=src)
        t.copy_(c)
    else:
        dist.broadcast(t, src=src)
    return t


class CellSumExchange:
    """The per-source-cell (sum area, sum clon, sum clat) of setup_conserve_interp, bit-identical for any number of ranks.

    conserve_interp.c:203-221 gathers the exchange cells of every rank and adds them to the per-cell accumulators one by one --
    "for the purpose of bitwise reproducing" -- output tile after output tile, rank after rank.  An all-reduce of per-rank
    partial sums adds the same terms in a different association for the cells that have exchange cells on two ranks (or in two
    output tiles), so instead ONE running total is handed along: every plan continues the sums where the previous one stopped
    (fg_plan_accumulate_cell_sums).  Across ranks only the cells present on more than one rank need the hand-over; they are
    passed from rank to rank by broadcasts of that short list, per output tile, in rank order.

    The object is the communication SCHEDULE of one decomposition: the list of shared cells depends only on the grids and the
    bands (found with one all-reduce of a presence mask: 0.3 % of the cells for 2 bands of C384), so a job that searches the
    same grids again (bench.py's timed steps) builds it once and calls run() per search.  run() issues no host synchronisation
    of its own when the plans sit on torch's current stream (XgridPlan.accumulate_cell_sums then only queues its kernel):
    search, hand-over and centroid pass are ordered by the stream."""

    def __init__(self, plans, device=0):
        import torch
        import torch.distributed as dist
        self.dev = device if isinstance(device, str) else f"cuda:{device}"      # ("cpu" with stand-in plans: the gloo test of this logic)
        self.on_gpu = self.dev.startswith("cuda")
        self.ncell = plans[0].ncells_in
        self.W = world_size()
        self.rank = dist.get_rank() if self.W > 1 else 0
        self.shared = self.idx3 = None
        self.nsh = 0
        if self.W == 1:
            return
        mine = torch.zeros(3 * self.ncell, dtype=torch.float64, device=self.dev)
        for p in plans:
            if p.nxgrid > 0:
                p.accumulate_cell_sums(mine)
        count = (mine[:self.ncell] != 0).to(torch.int32)                     # source cells with exchange cells on this rank
        _all_reduce(count)
        self.shared = torch.nonzero(count > 1).flatten().to(torch.int32)     # ... on more than one rank
        self.nsh = int(self.shared.numel())
        if self.nsh:
            idx = self.shared.long()
            self.idx3 = torch.cat([idx, idx + self.ncell, idx + 2 * self.ncell])
            self.scratch = torch.zeros(3 * self.ncell, dtype=torch.float64, device=self.dev)

    def run(self, plans, complete=True):
        """Returns the device tensor [3 * ncells_in].  complete=True: identical on every rank (a dense all-reduce fills in the
        cells of the other ranks, as conserve_interp.c's cell_in arrays are).  complete=False: exact for every source cell
        that has exchange cells on THIS rank -- all the centroid pass (fg_plan_finalize) reads -- and without the dense
        all-reduce (21 MB at C384)."""
        import torch
        total = torch.zeros(3 * self.ncell, dtype=torch.float64, device=self.dev)
        for p in plans:                                                        # output tile after output tile
            if p.nxgrid > 0:
                p.accumulate_cell_sums(total)
        if self.W > 1:
            run = None
            if self.nsh:
                run = torch.zeros(3 * self.nsh, dtype=torch.float64, device=self.dev)     # the running sums of the shared cells
                for p in plans:                                                # output tile after output tile ...
                    for r in range(self.W):                                    # ... rank after rank
                        if r == self.rank and p.nxgrid > 0:
                            self.scratch[self.idx3] = run
                            p.accumulate_cell_sums(self.scratch, self.shared)
                            run = self.scratch[self.idx3].contiguous()
                        _broadcast(run, r)
            if complete:
                if self.nsh:
                    total[self.idx3] = 0.0
                _all_reduce(total)                                             # every other cell: its one rank's value + zeros
            if self.nsh:
                total[self.idx3] = run
        return total


def ordered_cell_sums(plans, device=0, complete=True):
    """One-shot form of CellSumExchange (schedule + run); returns with the device idle."""
    import torch
    ex = CellSumExchange(plans, device)
    total = ex.run(plans, complete=complete)
    if ex.on_gpu:
        torch.cuda.synchronize(ex.dev)
    return total
