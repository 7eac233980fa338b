"""Host mirror of tools/fregrid/conserve_interp.c over the device-resident plan API.

``setup_conserve_interp`` / ``do_scalar_conserve_interp`` keep the reference's names, argument
order and meaning (conserve_interp.c:42, :507); the config structs are small Python stand-ins
for ``Grid_config`` / ``Interp_config`` / ``Field_config`` / ``Var_config``
(tools/libfrencutils/globals.h:67-216) holding only the members this path reads.

PyTorch is used for device buffers and (when a process group is initialised) for the one
collective of the path: the sum of the per-source-cell (area, clon, clat) partial sums over
ranks -- conserve_interp.c:203-221 does the same with mpp_gather + a serial sum.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import lib, check

# option bits, tools/libfrencutils/globals.h:46-61
CONSERVE_ORDER1 = 1
CONSERVE_ORDER2 = 2
TARGET = 16
READ = 256
WRITE = 512
CHECK_CONSERVE = 1024
LEGACY_CLIP = 2048
GREAT_CIRCLE = 4096
MONOTONIC = 16384
CELL_METHODS_MEAN = 0      # globals.h:64-65
CELL_METHODS_SUM = 1
MAXVAL = 1.0e20            # conserve_interp.c:36


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


@dataclass
class GridConfig:
    """Grid_config (globals.h:170-216): nx, ny and corner arrays lonc/latc [(ny+1),(nx+1)] radians.
    For an output grid split over ranks pass the rank's band (nxc, nyc and its corners), as
    get_output_grid_by_size does (fregrid_util.c:645-654)."""
    nx: int
    ny: int
    lonc: np.ndarray
    latc: np.ndarray
    cell_area: Optional[np.ndarray] = None     # filled by setup_conserve_interp (get_grid_area semantics)
    weight: Optional[np.ndarray] = None        # grid_in[].weight [ny, nx] (--weight_file/--weight_field)
    weight_exist: int = 0
    isc: int = 0                               # start of this rank's compute domain in the whole output tile
    jsc: int = 0                               # (grid_out[n].isc / jsc, fregrid_util.c:598-603)

    @property
    def nxc(self):
        return self.nx

    @property
    def nyc(self):
        return self.ny


@dataclass
class VarConfig:
    """Var_config (globals.h:67-100), members read by do_scalar_conserve_interp."""
    name: str = "var"
    interp_method: int = CONSERVE_ORDER1
    has_missing: int = 0
    missing: float = -MAXVAL
    cell_measures: int = 0                     # the variable carries its own cell area (field_in[].area)
    cell_methods: int = CELL_METHODS_MEAN
    area_missing: float = -MAXVAL
    use_volume: int = 0


@dataclass
class FieldConfig:
    """Field_config (globals.h:102-112) for ONE tile.  data is [nz, ny, nx] (order 1) or
    [nz, ny+2, nx+2] (order 2, halo 1); grad_x/grad_y [nz, ny, nx]; grad_mask [ny, nx] int."""
    data: Optional[np.ndarray] = None
    grad_x: Optional[np.ndarray] = None
    grad_y: Optional[np.ndarray] = None
    grad_mask: Optional[np.ndarray] = None
    area: Optional[np.ndarray] = None          # field_in[].area [ny, nx] of a cell_measures variable
    var: List[VarConfig] = field(default_factory=lambda: [VarConfig()])


class XgridPlan:
    """RAII wrapper of an ``fg_plan`` (include/fregrid_hip.h)."""

    def __init__(self, handle, order, device, great_circle=False):
        self._h = C.c_void_p(handle)
        self.order = order
        self.device = device
        self.great_circle = great_circle

    # -- construction -------------------------------------------------------------------
    @classmethod
    def create(cls, order, grids_in, grid_out, masks=None, device=0):
        """Search with host corner arrays (copied to the device)."""
        _lib.require_gpu()
        L = lib()
        nt = len(grids_in)
        nx = (C.c_int * nt)(*[g.nx for g in grids_in])
        ny = (C.c_int * nt)(*[g.ny for g in grids_in])
        keep = []
        dpt = C.POINTER(C.c_double)

        def arr(a, n):
            a = _f64(a).reshape(-1)
            assert a.size == n, (a.size, n)
            keep.append(a)
            return _dp(a)

        lon = (dpt * nt)(*[arr(g.lonc, (g.nx + 1) * (g.ny + 1)) for g in grids_in])
        lat = (dpt * nt)(*[arr(g.latc, (g.nx + 1) * (g.ny + 1)) for g in grids_in])
        if masks is None:
            msk = None
        else:
            msk = (dpt * nt)(*[arr(m, g.nx * g.ny) if m is not None else dpt() for m, g in zip(masks, grids_in)])
        lo = arr(grid_out.lonc, (grid_out.nx + 1) * (grid_out.ny + 1))
        la = arr(grid_out.latc, (grid_out.nx + 1) * (grid_out.ny + 1))
        h = C.c_void_p()
        check(L.fg_plan_create(order, nt, nx, ny, lon, lat, msk, grid_out.nx, grid_out.ny, lo, la, device, C.byref(h)))
        return cls(h.value, order, device)

    @classmethod
    def create_dev(cls, order, nx_in, ny_in, lon_in_t, lat_in_t, nx_out, ny_out, lon_out_t, lat_out_t,
                   mean_dlat=0.0, mean_dlon=0.0, device=0, stream=None, masks_t=None):
        """Search with corner arrays already on the device (torch float64 CUDA tensors)."""
        L = lib()
        nt = len(nx_in)
        nx = (C.c_int * nt)(*nx_in)
        ny = (C.c_int * nt)(*ny_in)
        lon = (C.c_void_p * nt)(*[t.data_ptr() for t in lon_in_t])
        lat = (C.c_void_p * nt)(*[t.data_ptr() for t in lat_in_t])
        msk = None
        if masks_t is not None:
            msk = (C.c_void_p * nt)(*[(t.data_ptr() if t is not None else None) for t in masks_t])
        h = C.c_void_p()
        use = 0 if stream is None else 1
        sptr = C.c_void_p(0 if stream is None else int(stream))
        check(L.fg_plan_create_dev(order, nt, nx, ny, lon, lat, msk, nx_out, ny_out,
                                   C.c_void_p(lon_out_t.data_ptr()), C.c_void_p(lat_out_t.data_ptr()),
                                   float(mean_dlat), float(mean_dlon), device, sptr, use, C.byref(h)))
        return cls(h.value, order, device)

    @classmethod
    def create_great_circle(cls, grids_in, grid_out, masks=None, device=0):
        """Great-circle search (create_xgrid_great_circle semantics) with host corner arrays; first order."""
        _lib.require_gpu()
        L = lib()
        nt = len(grids_in)
        nx = (C.c_int * nt)(*[g.nx for g in grids_in])
        ny = (C.c_int * nt)(*[g.ny for g in grids_in])
        keep = []
        dpt = C.POINTER(C.c_double)

        def arr(a, n):
            a = _f64(a).reshape(-1)
            assert a.size == n, (a.size, n)
            keep.append(a)
            return _dp(a)

        lon = (dpt * nt)(*[arr(g.lonc, (g.nx + 1) * (g.ny + 1)) for g in grids_in])
        lat = (dpt * nt)(*[arr(g.latc, (g.nx + 1) * (g.ny + 1)) for g in grids_in])
        msk = None
        if masks is not None:
            msk = (dpt * nt)(*[arr(m, g.nx * g.ny) if m is not None else dpt() for m, g in zip(masks, grids_in)])
        lo = arr(grid_out.lonc, (grid_out.nx + 1) * (grid_out.ny + 1))
        la = arr(grid_out.latc, (grid_out.nx + 1) * (grid_out.ny + 1))
        h = C.c_void_p()
        check(L.fg_plan_create_great_circle(nt, nx, ny, lon, lat, msk, grid_out.nx, grid_out.ny, lo, la, device, C.byref(h)))
        return cls(h.value, 1, device, True)

    @classmethod
    def create_great_circle_dev(cls, nx_in, ny_in, xyz_in_t, nx_out, ny_out, xyz_out_t, mean_dlat=0.0, mean_dlon=0.0,
                                device=0, stream=None, masks_t=None):
        """Great-circle search on unit vectors already on the device: xyz_in_t[m] = (x, y, z) torch tensors."""
        L = lib()
        nt = len(nx_in)
        nx = (C.c_int * nt)(*nx_in)
        ny = (C.c_int * nt)(*ny_in)
        xs = (C.c_void_p * nt)(*[t[0].data_ptr() for t in xyz_in_t])
        ys = (C.c_void_p * nt)(*[t[1].data_ptr() for t in xyz_in_t])
        zs = (C.c_void_p * nt)(*[t[2].data_ptr() for t in xyz_in_t])
        msk = None
        if masks_t is not None:
            msk = (C.c_void_p * nt)(*[(t.data_ptr() if t is not None else None) for t in masks_t])
        h = C.c_void_p()
        use = 0 if stream is None else 1
        sptr = C.c_void_p(0 if stream is None else int(stream))
        check(L.fg_plan_create_great_circle_dev(nt, nx, ny, xs, ys, zs, msk, nx_out, ny_out,
                                                C.c_void_p(xyz_out_t[0].data_ptr()), C.c_void_p(xyz_out_t[1].data_ptr()),
                                                C.c_void_p(xyz_out_t[2].data_ptr()), float(mean_dlat), float(mean_dlon),
                                                device, sptr, use, C.byref(h)))
        return cls(h.value, 1, device, True)

    @classmethod
    def create_empty(cls, order, nx_in, ny_in, nx_out, ny_out, device=0):
        _lib.require_gpu()
        nt = len(nx_in)
        h = C.c_void_p()
        check(lib().fg_plan_create_empty(order, nt, (C.c_int * nt)(*nx_in), (C.c_int * nt)(*ny_in),
                                         nx_out, ny_out, device, C.byref(h)))
        return cls(h.value, order, device)

    def destroy(self):
        if self._h is not None and self._h.value:
            lib().fg_plan_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- accessors ----------------------------------------------------------------------
    @property
    def nxgrid(self):
        return int(lib().fg_plan_nxgrid(self._h))

    @property
    def ncells_in(self):
        return int(lib().fg_plan_ncells_in(self._h))

    def cell_sums_ptr(self):
        return lib().fg_plan_cell_sums_dev(self._h)

    def copy_cell_sums(self, dst_t):
        check(lib().fg_plan_copy_cell_sums(self._h, C.c_void_p(dst_t.data_ptr())))

    def accumulate_cell_sums(self, total_t, cells_t=None):
        """total_t[3, ncells_in] += this plan's exchange cells, one by one in exchange-cell order, continuing from the values
        already there (conserve_interp.c:203-221); cells_t: int32 device tensor restricting the update to those source cells.

        The kernel runs on the PLAN's stream while total_t / cells_t are usually the product of torch ops queued on torch's
        current stream (a zero fill, an index_put of the sums received from another rank, the wait on a collective).  If the
        plan IS on torch's current stream (create_dev(stream=...) / set_stream) stream order does it all and the call only
        queues the kernel; otherwise torch's stream is drained first -- the kernel would read stale values -- and the C call
        returns with the plan's stream drained, so torch ops issued afterwards see the result."""
        args = (self._h, C.c_void_p(total_t.data_ptr()), C.c_void_p(cells_t.data_ptr()) if cells_t is not None else None,
                int(cells_t.numel()) if cells_t is not None else 0)
        if getattr(total_t, "is_cuda", False):
            cur = _torch().cuda.current_stream(total_t.device)
            if int(self.stream() or 0) == int(cur.cuda_stream):      # (a plan with a stream of its own never reports 0)
                check(lib().fg_plan_accumulate_cell_sums_async(*args))
                return
            cur.synchronize()
        check(lib().fg_plan_accumulate_cell_sums(*args))

    def stats(self):
        s = (C.c_long * 10)()
        check(lib().fg_plan_stats(self._h, s, 10))
        names = ["pairs", "nonempty", "nxgrid", "borderline", "bins", "bin_entries", "deferred", "heavy", "below", "exact_mode"]
        return dict(zip(names, [int(v) for v in s]))

    PHASES = ["cell_struct", "bins", "candidates", "clip_quad", "clip_general", "compact", "rows",
              "search_total", "finalize", "apply"]

    def phase_ms(self):
        ms = (C.c_float * 10)()
        check(lib().fg_plan_phase_ms(self._h, ms, 10))
        return dict(zip(self.PHASES, [float(v) for v in ms]))

    def finalize(self, total_sums_ptr=None):
        check(lib().fg_plan_finalize(self._h, C.c_void_p(total_sums_ptr or 0)))

    def sync(self):
        check(lib().fg_plan_sync(self._h))

    def stream(self):
        return lib().fg_plan_stream(self._h)

    def set_stream(self, stream):
        check(lib().fg_plan_set_stream(self._h, C.c_void_p(int(stream))))

    def get_xgrid(self):
        """dict with t_in,i_in,j_in,i_out,j_out (int32), area and c1/c2 (order 2)."""
        n = self.nxgrid
        ints = {k: np.empty(n, dtype=np.int32) for k in ("t_in", "i_in", "j_in", "i_out", "j_out")}
        area = np.empty(n, dtype=np.float64)
        c1 = np.empty(n, dtype=np.float64) if self.order == 2 else None
        c2 = np.empty(n, dtype=np.float64) if self.order == 2 else None
        check(lib().fg_plan_get_xgrid(self._h, _ip(ints["t_in"]), _ip(ints["i_in"]), _ip(ints["j_in"]),
                                      _ip(ints["i_out"]), _ip(ints["j_out"]), _dp(area), _dp(c1), _dp(c2)))
        out = dict(ints)
        out["area"] = area
        if self.order == 2:
            out["c1"], out["c2"] = c1, c2
        return out

    def get_polygons(self, maxv=16):
        """The clipped polygon of every exchange cell (fg_plan_get_polygons): dict n [nxgrid] and, for a legacy plan, lon / lat
        [nxgrid, maxv]; for a great-circle plan x / y / z."""
        nx = self.nxgrid
        n = np.zeros(nx, dtype=np.int32)
        v = [np.zeros((nx, maxv)) for _ in range(3)]
        check(lib().fg_plan_get_polygons(self._h, maxv, _ip(n), _dp(v[0]), _dp(v[1]), _dp(v[2])))
        if self.great_circle:
            return {"n": n, "x": v[0], "y": v[1], "z": v[2]}
        return {"n": n, "lon": v[0], "lat": v[1]}

    @classmethod
    def create_polylist(cls, order, n, lon, lat, lon_avg, area_ref, grid_out, device=0):
        """A search whose source cells are polygons of <= 8 vertices (fg_plan_create_polylist): n [npoly], lon / lat [npoly, 8],
        lon_avg / area_ref [npoly]."""
        _lib.require_gpu()
        n = np.ascontiguousarray(n, dtype=np.int32)
        lon, lat = _f64(lon), _f64(lat)
        assert lon.shape == (n.size, 8) == lat.shape
        lon_avg, area_ref = _f64(lon_avg), _f64(area_ref)
        lo, la = _f64(grid_out.lonc).reshape(-1), _f64(grid_out.latc).reshape(-1)
        h = C.c_void_p()
        check(lib().fg_plan_create_polylist(order, n.size, _ip(n), _dp(lon), _dp(lat), _dp(lon_avg), _dp(area_ref), grid_out.nx, grid_out.ny,
                                            _dp(lo), _dp(la), device, C.byref(h)))
        return cls(h.value, order, device)

    def get_cell_area(self, ncells_out):
        a_in = np.empty(self.ncells_in, dtype=np.float64)
        a_out = np.empty(ncells_out, dtype=np.float64)
        check(lib().fg_plan_get_cell_area(self._h, _dp(a_in), _dp(a_out)))
        return a_in, a_out

    def get_cell_struct(self, which, ncells):
        d = {k: np.empty(ncells, dtype=np.float64) for k in ("lat_min", "lat_max", "lon_min", "lon_max", "lon_avg")}
        nv = np.empty(ncells, dtype=np.int32)
        vlon = np.empty((ncells, 8), dtype=np.float64)
        vlat = np.empty((ncells, 8), dtype=np.float64)
        check(lib().fg_plan_get_cell_struct(self._h, which, _dp(d["lat_min"]), _dp(d["lat_max"]), _dp(d["lon_min"]),
                                            _dp(d["lon_max"]), _dp(d["lon_avg"]), _ip(nv), _dp(vlon), _dp(vlat)))
        d.update(nvert=nv, vlon=vlon, vlat=vlat)
        return d

    def set_xgrid(self, t_in, i_in, j_in, i_out, j_out, area, di_in=None, dj_in=None):
        a = [np.ascontiguousarray(v, dtype=np.int32) for v in (t_in, i_in, j_in, i_out, j_out)]
        area = _f64(area)
        di = _f64(di_in) if di_in is not None else None
        dj = _f64(dj_in) if dj_in is not None else None
        check(lib().fg_plan_set_xgrid(self._h, area.size, *[_ip(v) for v in a], _dp(area), _dp(di), _dp(dj)))

    def apply(self, data_t, out_t, nz=1, grad_x_t=None, grad_y_t=None, grad_mask_t=None,
              has_missing=False, missing=-MAXVAL, want_gsum=False):
        """Sweep on device tensors (torch float64 / int32 CUDA tensors)."""
        g = C.c_double(0.0)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        check(lib().fg_plan_apply(self._h, ptr(data_t), ptr(grad_x_t), ptr(grad_y_t), ptr(grad_mask_t),
                                  1 if has_missing else 0, float(missing), nz, ptr(out_t),
                                  C.byref(g) if want_gsum else None))
        return g.value if want_gsum else None


    def apply_ex(self, data_t, out_t, nz=1, grad_x_t=None, grad_y_t=None, grad_mask_t=None, has_missing=False,
                 missing=-MAXVAL, weight_t=None, cell_methods_sum=False, field_area_t=None, area_missing=-MAXVAL,
                 cell_area_in_t=None, cell_area_out_t=None, monotonic=False, want_gsum=False, minmax_hook=None):
        """The sweep with every do_scalar_conserve_interp option (fg_plan_apply_ex).  ``minmax_hook(plan)``
        is called between the two halves of the monotone sweep (all-reduce point, :672-677)."""
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        addr = lambda t: t.data_ptr() if t is not None else None
        o = _lib.ApplyOpts(1 if has_missing else 0, float(missing), addr(weight_t), 1 if cell_methods_sum else 0,
                           addr(field_area_t), float(area_missing), addr(cell_area_in_t), addr(cell_area_out_t),
                           1 if monotonic else 0)
        g = C.c_double(0.0)
        gp = C.byref(g) if want_gsum else None
        L = lib()
        if monotonic and self.order == 2 and minmax_hook is not None:
            check(L.fg_plan_mono_begin(self._h, C.byref(o), ptr(data_t), ptr(grad_x_t), ptr(grad_y_t), ptr(grad_mask_t)))
            minmax_hook(self)
            check(L.fg_plan_mono_end(self._h, C.byref(o), ptr(data_t), ptr(out_t), gp))
        else:
            check(L.fg_plan_apply_ex(self._h, C.byref(o), ptr(data_t), ptr(grad_x_t), ptr(grad_y_t), ptr(grad_mask_t),
                                     nz, ptr(out_t), gp))
        return g.value if want_gsum else None

    def mono_copy_minmax(self, to_plan, fmin_t, fmax_t):
        check(lib().fg_plan_mono_copy_minmax(self._h, 1 if to_plan else 0, C.c_void_p(fmin_t.data_ptr()),
                                              C.c_void_p(fmax_t.data_ptr())))

    def apply_interleaved(self, nb, data_il_t, out_il_t, grad_x_il_t=None, grad_y_il_t=None, want_gsum=False):
        """Sweep on interleaved device tensors: data [F, nb], grads [ncells_in, nb], out [ndst, nb]."""
        g = C.c_double(0.0)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        check(lib().fg_plan_apply_interleaved(self._h, nb, ptr(data_il_t), ptr(grad_x_il_t), ptr(grad_y_il_t),
                                              ptr(out_il_t), C.byref(g) if want_gsum else None))
        return g.value if want_gsum else None

    def apply_records(self, nz, rec_t, out_t, want_gsum=False):
        """Order-2 sweep of nz <= 8 levels on the records C2lPrep.gradient_records wrote ([ncells_in, 3, nb] device tensor);
        out [nz, ndst] level-major.  Bit-identical to apply() on the level-major gradients."""
        g = C.c_double(0.0)
        check(lib().fg_plan_apply_records(self._h, nz, C.c_void_p(rec_t.data_ptr()), C.c_void_p(out_t.data_ptr()),
                                          C.byref(g) if want_gsum else None))
        return g.value if want_gsum else None


@dataclass
class InterpConfig:
    """Interp_config (globals.h:144-158).  The arrays are host copies of the plan's exchange
    cells; ``plan`` keeps them resident in HBM for do_scalar_conserve_interp."""
    nxgrid: int = 0
    i_in: Optional[np.ndarray] = None
    j_in: Optional[np.ndarray] = None
    i_out: Optional[np.ndarray] = None
    j_out: Optional[np.ndarray] = None
    t_in: Optional[np.ndarray] = None
    di_in: Optional[np.ndarray] = None
    dj_in: Optional[np.ndarray] = None
    area: Optional[np.ndarray] = None
    plan: Optional[XgridPlan] = None
    remap_file: Optional[str] = None           # Interp_config.remap_file (globals.h:155); used with READ / WRITE
    file_exist: int = 0


def _torch():
    import torch
    return torch


def get_input_output_cell_area(ntiles_in, grid_in, ntiles_out, grid_out, opcode):
    """fregrid_util.c:363-408: grid_in[].cell_area and grid_out[].cell_area from get_grid_area, or get_grid_great_circle_area
    with GREAT_CIRCLE (the B1 entry points of the library: device work, host arrays in and out)."""
    from . import get_grid_area, get_grid_great_circle_area
    fn = get_grid_great_circle_area if (opcode & GREAT_CIRCLE) else get_grid_area
    for g in list(grid_in[:ntiles_in]) + list(grid_out[:ntiles_out]):
        g.cell_area = fn(g.nx, g.ny, g.lonc, g.latc)


def setup_conserve_interp(ntiles_in, grid_in, ntiles_out, grid_out, interp, opcode, device=0, fetch=True, cull=None):
    """conserve_interp.c:42, compute branch (:127-358).  Fills interp[n] for each output tile.

    With torch.distributed initialised (world_size > 1) each rank passes its own band of the output grid (grid_out[n] with
    isc / jsc set), searches it with the source cells culled to the band (cull=None: on exactly then) and the per-source-cell
    sums of the cells present on several ranks are handed from rank to rank in the reference's order before the centroid
    pass (parallel.CellSumExchange, :203-221).  GREAT_CIRCLE: first order, no exchange at all; WRITE gathers on the root.
    """
    order = 2 if (opcode & CONSERVE_ORDER2) else 1
    if not (opcode & (CONSERVE_ORDER1 | CONSERVE_ORDER2)):
        raise ValueError("conserve_interp: interp_method should be CONSERVE_ORDER1 or CONSERVE_ORDER2")
    torch = _torch()
    if opcode & READ:                                                  # conserve_interp.c:62-126
        from .remap_file import read_remap_file
        for n in range(ntiles_out):
            ic = interp[n]
            if not ic.file_exist:
                continue
            x = read_remap_file(ic.remap_file, order)
            g = grid_out[n]
            keep = ((x["i_out"] >= g.isc) & (x["i_out"] <= g.isc + g.nx - 1) &
                    (x["j_out"] >= g.jsc) & (x["j_out"] <= g.jsc + g.ny - 1))          # :93-97
            sel = {k: v[keep] for k, v in x.items()}
            sel["i_out"] = sel["i_out"] - g.isc
            sel["j_out"] = sel["j_out"] - g.jsc
            plan = XgridPlan.create_empty(order, [gi.nx for gi in grid_in[:ntiles_in]], [gi.ny for gi in grid_in[:ntiles_in]],
                                          g.nx, g.ny, device=device)
            plan.set_xgrid(sel["t_in"], sel["i_in"], sel["j_in"], sel["i_out"], sel["j_out"], sel["area"],
                           sel.get("di_in"), sel.get("dj_in"))
            ic.plan, ic.nxgrid = plan, int(keep.sum())
            ic.t_in, ic.i_in, ic.j_in, ic.i_out, ic.j_out, ic.area = (sel["t_in"], sel["i_in"], sel["j_in"], sel["i_out"],
                                                                       sel["j_out"], sel["area"])
            ic.di_in, ic.dj_in = sel.get("di_in"), sel.get("dj_in")
        print("NOTE: Finish reading index and weight for conservative interpolation from file.")   # :125
        return interp
    plans = []
    great_circle = bool(opcode & GREAT_CIRCLE)
    if great_circle and order != 1:
        raise ValueError("fregrid: when clip_method is 'conserve_great_circle', interp_method must be 'conserve_order1'")  # fregrid.c:763
    # A rank of a banded job (fregrid_parallel: grid_out[n] is this rank's latitude band, fregrid_util.c:592-603) meets a fraction
    # of the source cells: the search then skips the others when it builds the per-cell records (fg_set_search_cull) -- the
    # counterpart of the reference's jstart / jend row trim (conserve_interp.c:169-184), for the great-circle search too.
    from .parallel import world_size
    cull = world_size() > 1 if cull is None else bool(cull)
    lib().fg_set_search_cull(1 if cull else 0)
    # first order needs no sums from anybody, second order with one destination tile on one rank has them in its own plan: the
    # search then queues its finalize work itself, without a host round trip in between (fg_set_search_finalize)
    fused = (order == 1 and not great_circle) or (order == 2 and ntiles_out == 1 and world_size() == 1)
    lib().fg_set_search_finalize(1 if fused else 0)
    try:
        for n in range(ntiles_out):
            if great_circle:                                           # conserve_interp.c:164-168 (whole tiles, no row trim)
                plans.append(XgridPlan.create_great_circle(grid_in[:ntiles_in], grid_out[n], device=device))
            else:
                plans.append(XgridPlan.create(order, grid_in[:ntiles_in], grid_out[n], device=device))
    finally:
        lib().fg_set_search_cull(0)
        lib().fg_set_search_finalize(0)
    if cull:
        # a culling search holds no area for the source cells it skipped
        get_input_output_cell_area(ntiles_in, grid_in, ntiles_out, grid_out, opcode)
    else:
        # cell areas (fregrid_util.c:363-408) come for free from the search
        for n in range(ntiles_out):
            a_in, a_out = plans[n].get_cell_area(grid_out[n].nx * grid_out[n].ny)
            grid_out[n].cell_area = a_out
            if n == 0:
                off = 0
                for g in grid_in[:ntiles_in]:
                    g.cell_area = a_in[off:off + g.nx * g.ny].copy()
                    off += g.nx * g.ny
    if order == 2 and fused:
        plans[0].finalize(None)                    # (done by its search; returns at once)
    elif order == 2:
        from .parallel import ordered_cell_sums
        total = ordered_cell_sums(plans, device)   # the one exchange step of the path, in the reference's order
        for p in plans:
            p.finalize(total.data_ptr())
    else:
        for p in plans:
            p.finalize(None)
    for n in range(ntiles_out):
        ic = interp[n]
        ic.plan = plans[n]
        ic.nxgrid = plans[n].nxgrid
        if fetch:
            x = plans[n].get_xgrid()
            ic.t_in, ic.i_in, ic.j_in, ic.i_out, ic.j_out, ic.area = (x["t_in"], x["i_in"], x["j_in"], x["i_out"],
                                                                       x["j_out"], x["area"])
            if order == 2:
                ic.di_in, ic.dj_in = x["c1"], x["c2"]
    if opcode & WRITE:                                                 # conserve_interp.c:368-445
        for n in range(ntiles_out):
            ic = interp[n]
            if not ic.remap_file:
                continue
            if not fetch and ic.nxgrid > 0:
                x = plans[n].get_xgrid()
                ic.t_in, ic.i_in, ic.j_in, ic.i_out, ic.j_out, ic.area = (x["t_in"], x["i_in"], x["j_in"], x["i_out"],
                                                                           x["j_out"], x["area"])
                if order == 2:
                    ic.di_in, ic.dj_in = x["c1"], x["c2"]
            write_remap_gathered(ic, grid_out[n], order)
    print("NOTE: done calculating index and weight for conservative interpolation")   # :446
    return interp


def write_remap_gathered(ic, grid_out_n, order):
    """The WRITE branch for one output tile (conserve_interp.c:368-445): mpp_sum_int of nxgrid, the exchange cells of every rank
    gathered on the root in rank order (mpp_gather_field_int / _double; output indices made global with isc / jsc BEFORE the
    gather, :407,:416), one file written by the root.  Every rank must call it (it is a collective when a process group is
    initialised); returns the global nxgrid.  Host arrays only: no device work."""
    from .parallel import world_size
    from .remap_file import write_remap_file
    n_loc = int(ic.nxgrid)
    z_i, z_d = np.zeros(0, dtype=np.int32), np.zeros(0)
    part = {"t_in": ic.t_in if n_loc else z_i, "i_in": ic.i_in if n_loc else z_i, "j_in": ic.j_in if n_loc else z_i,
            "i_out": (np.asarray(ic.i_out) + grid_out_n.isc) if n_loc else z_i,
            "j_out": (np.asarray(ic.j_out) + grid_out_n.jsc) if n_loc else z_i,
            "area": ic.area if n_loc else z_d}
    if order == 2:
        part["di_in"], part["dj_in"] = (ic.di_in if n_loc else z_d), (ic.dj_in if n_loc else z_d)
    rank = 0
    if world_size() > 1:
        import torch.distributed as dist
        rank = dist.get_rank()
        parts = [None] * world_size()
        dist.all_gather_object(parts, {k: np.ascontiguousarray(v) for k, v in part.items()})
        part = {k: np.concatenate([p[k] for p in parts]) for k in part}
    n_glob = int(part["area"].size)
    if rank == 0 and n_glob > 0:
        write_remap_file(ic.remap_file, order, part["t_in"], part["i_in"], part["j_in"], part["i_out"], part["j_out"], part["area"],
                         part.get("di_in"), part.get("dj_in"), 0, 0)
    if world_size() > 1:
        import torch.distributed as dist
        dist.barrier()                                                # the file exists when any rank returns
    return n_glob


def pack_field(order, field_in, ntiles_in, nz, key="data"):
    """Concatenate per-tile arrays into the device layout [nz][tiles back to back]."""
    lv = []
    for k in range(nz):
        parts = []
        for t in range(ntiles_in):
            a = getattr(field_in[t], key)
            a = np.asarray(a)
            a = a.reshape(nz, -1) if a.ndim != 2 or a.shape[0] != nz else a
            parts.append(a[k].reshape(-1))
        lv.append(np.concatenate(parts))
    return np.ascontiguousarray(np.stack(lv))


def do_scalar_conserve_interp(interp, varid, ntiles_in, grid_in, ntiles_out, grid_out, field_in, field_out,
                              opcode, nz, device=0):
    """conserve_interp.c:507-910.  The plain branch (no weight / cell_measures / cell_methods=sum / monotonic /
    target_grid) runs the bandwidth kernels (fg_plan_apply); any of those options routes to fg_plan_apply_ex."""
    torch = _torch()
    var = field_in[0].var[varid]
    order = 2 if var.interp_method == CONSERVE_ORDER2 else 1
    monotonic = bool(opcode & MONOTONIC) if order == 2 else False      # :525-531
    has_missing = int(var.has_missing)
    missing = var.missing if has_missing else -MAXVAL
    weight_exist = int(grid_in[0].weight_exist)
    cell_measures, cell_methods = int(var.cell_measures), int(var.cell_methods)
    target_grid = bool(opcode & TARGET) and not var.use_volume          # :535-536
    if nz > 1 and has_missing:
        raise ValueError("conserve_interp: has_missing should be false when nz > 1")
    if nz > 1 and cell_measures:
        raise ValueError("conserve_interp: cell_measures should be false when nz > 1")
    if nz > 1 and cell_methods == CELL_METHODS_SUM:
        raise ValueError("conserve_interp: cell_methods should not be sum when nz > 1")
    dev = f"cuda:{device}"
    data = torch.from_numpy(pack_field(order, field_in, ntiles_in, nz, "data").astype(np.float64)).to(dev)
    gx = gy = gm = None
    if order == 2:
        gx = torch.from_numpy(pack_field(order, field_in, ntiles_in, nz, "grad_x").astype(np.float64)).to(dev)
        gy = torch.from_numpy(pack_field(order, field_in, ntiles_in, nz, "grad_y").astype(np.float64)).to(dev)
        if has_missing or (monotonic and field_in[0].grad_mask is not None):
            gm = torch.from_numpy(np.concatenate([np.asarray(field_in[t].grad_mask, dtype=np.int32).reshape(-1)
                                                  for t in range(ntiles_in)])).to(dev)
    extended = weight_exist or cell_measures or cell_methods == CELL_METHODS_SUM or target_grid or monotonic

    def flat(get):
        return torch.from_numpy(np.concatenate([_f64(get(t)).reshape(-1) for t in range(ntiles_in)])).to(dev)

    w_t = fa_t = ca_t = None
    if extended:
        if weight_exist:
            w_t = flat(lambda t: grid_in[t].weight)
        if cell_measures:
            fa_t = flat(lambda t: field_in[t].area)
        if cell_measures or cell_methods == CELL_METHODS_SUM:
            ca_t = flat(lambda t: grid_in[t].cell_area)

    def minmax_hook(plan):                                             # mpp_min_double / mpp_max_double, :672-677
        from .parallel import allreduce_minmax, world_size
        if world_size() > 1:
            fmin = torch.empty(plan.ncells_in, dtype=torch.float64, device=dev)
            fmax = torch.empty_like(fmin)
            plan.mono_copy_minmax(False, fmin, fmax)
            allreduce_minmax(fmin, fmax)
            torch.cuda.synchronize(device)
            plan.mono_copy_minmax(True, fmin, fmax)

    gsum_out = 0.0
    for m in range(ntiles_out):
        nx2, ny2 = grid_out[m].nxc, grid_out[m].nyc
        out = torch.empty(nz * nx2 * ny2, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(device)
        want = bool(opcode & CHECK_CONSERVE)
        if extended:
            cao_t = torch.from_numpy(_f64(grid_out[m].cell_area).reshape(-1)).to(dev) if target_grid else None
            torch.cuda.synchronize(device)
            g = interp[m].plan.apply_ex(data, out, nz=nz, grad_x_t=gx, grad_y_t=gy, grad_mask_t=gm,
                                        has_missing=bool(has_missing), missing=missing, weight_t=w_t,
                                        cell_methods_sum=cell_methods == CELL_METHODS_SUM, field_area_t=fa_t,
                                        area_missing=var.area_missing, cell_area_in_t=ca_t, cell_area_out_t=cao_t,
                                        monotonic=monotonic, want_gsum=want, minmax_hook=minmax_hook)
        else:
            g = interp[m].plan.apply(data, out, nz=nz, grad_x_t=gx, grad_y_t=gy, grad_mask_t=gm,
                                     has_missing=bool(has_missing), missing=missing, want_gsum=want)
        interp[m].plan.sync()
        field_out[m].data = out.cpu().numpy().reshape(nz, ny2, nx2)
        if g is not None:
            gsum_out += g
    if opcode & CHECK_CONSERVE:                                       # :874-907
        halo = 1 if order == 2 else 0
        gsum_in = 0.0
        for n in range(ntiles_in):
            nx1, ny1 = grid_in[n].nx, grid_in[n].ny
            d = np.asarray(field_in[n].data, dtype=np.float64).reshape(nz, ny1 + 2 * halo, nx1 + 2 * halo)
            d = d[:, halo:halo + ny1, halo:halo + nx1]
            if cell_measures:
                fa = np.asarray(field_in[n].area).reshape(ny1, nx1)
                gsum_in += float(np.sum(np.where(d[0] != missing, d[0] * fa, 0.0)))
            elif cell_methods == CELL_METHODS_SUM:
                gsum_in += float(np.sum(np.where(d[0] != missing, d[0], 0.0)))
            else:
                ca = np.asarray(grid_in[n].cell_area).reshape(ny1, nx1)
                gsum_in += float(np.sum(np.where(d != missing, d * ca[None], 0.0)))
        from .parallel import allreduce_scalar_sum
        gsum_out = allreduce_scalar_sum(gsum_out, dev)                 # mpp_sum_double, :902
        print("the flux(data*area) sum of %s: input = %g, output = %g, diff = %g. "
              % (var.name, gsum_in, gsum_out, gsum_out - gsum_in))
        return gsum_in, gsum_out
    return None
