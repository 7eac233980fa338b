"""fre-nctools_amd -- MI355X-native conservative regridding behind the FRE-NCtools ABI.

Host-side Python mirror of the reference interface for the conservative-regrid hot path.
All arithmetic runs in ``libfregrid_hip.so`` (hand-written HIP for gfx950, built from
``csrc/``); this module only binds its C ABI (``include/fregrid_hip.h``) with ctypes and
mirrors the reference's call signatures:

* ``create_xgrid_2dx2d_order1/2``, ``get_grid_area``, ``get_maxxgrid``, ``conserve_interp``
  -- tools/libfrencutils/create_xgrid.c:45,66,621,893 and interp.c:262
* ``setup_conserve_interp`` / ``do_scalar_conserve_interp``
  -- tools/fregrid/conserve_interp.c:42,507 (compute / READ / WRITE branches; every branch of the sweep: missing values,
     weight field, cell_methods = sum, cell_measures, --target_grid, the monotone limiter)

There is NO CPU fallback: if the shared library is missing, or no HIP device is visible
when a compute entry point is called, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import FregridHipError, lib, lib_path  # noqa: F401
from .grids import gnomonic_ed_corners, gnomonic_ed_grid, latlon_corners, tripolar_corners  # noqa: F401
from .c2l import C2lPrep, find_contacts, c2l_grid_info, halo_map  # noqa: F401
from .remap_file import write_remap_file, read_remap_file  # noqa: F401
from .field_io import NcFile, Sweep, HostBuffer, read_field_levels  # noqa: F401
from . import field_io  # noqa: F401
from .coupler import coupler_xgrid  # noqa: F401
from .parallel import (band_rows, row_cost, allreduce_cell_sums, allreduce_scalar_sum, allreduce_minmax,  # noqa: F401
                       boundary_source_cells, allreduce_cell_sums_sparse, ordered_cell_sums, CellSumExchange)
from .conserve_interp import (  # noqa: F401
    CONSERVE_ORDER1, CONSERVE_ORDER2, CHECK_CONSERVE, READ, WRITE, TARGET, MONOTONIC, GREAT_CIRCLE, LEGACY_CLIP, CELL_METHODS_MEAN, CELL_METHODS_SUM,
    GridConfig, InterpConfig, FieldConfig, VarConfig, XgridPlan,
    setup_conserve_interp, do_scalar_conserve_interp, write_remap_gathered,
)

__all__ = [
    "create_xgrid_2dx2d_order1", "create_xgrid_2dx2d_order2", "get_grid_area", "get_maxxgrid",
    "conserve_interp", "setup_conserve_interp", "do_scalar_conserve_interp",
    "GridConfig", "InterpConfig", "FieldConfig", "VarConfig", "XgridPlan",
    "gnomonic_ed_corners", "latlon_corners", "lib", "lib_path", "FregridHipError",
]


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def get_maxxgrid():
    """create_xgrid.c:45"""
    return int(lib().get_maxxgrid())


def get_grid_area(nlon, nlat, lon, lat):
    """create_xgrid.c:66 -- returns area[nlat*nlon] (m^2).  Fatal errors exit like the reference."""
    _lib.require_gpu()
    lon, lat = _f64(lon), _f64(lat)
    assert lon.size == (nlon + 1) * (nlat + 1) and lat.size == lon.size
    area = np.empty(nlon * nlat, dtype=np.float64)
    lib().get_grid_area(C.byref(C.c_int(nlon)), C.byref(C.c_int(nlat)), _dp(lon), _dp(lat), _dp(area))
    return area


def _create_xgrid(order, nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in):
    _lib.require_gpu()
    lon_in, lat_in, lon_out, lat_out = _f64(lon_in), _f64(lat_in), _f64(lon_out), _f64(lat_out)
    assert lon_in.size == (nlon_in + 1) * (nlat_in + 1) == lat_in.size
    assert lon_out.size == (nlon_out + 1) * (nlat_out + 1) == lat_out.size
    mask = _f64(np.ones(nlon_in * nlat_in) if mask_in is None else mask_in)
    assert mask.size == nlon_in * nlat_in
    cap = get_maxxgrid()                       # caller-allocated MAXXGRID arrays, as in the reference
    i_in, j_in, i_out, j_out = (np.empty(cap, dtype=np.int32) for _ in range(4))
    area = np.empty(cap, dtype=np.float64)
    args = [C.byref(C.c_int(nlon_in)), C.byref(C.c_int(nlat_in)), C.byref(C.c_int(nlon_out)),
            C.byref(C.c_int(nlat_out)), _dp(lon_in), _dp(lat_in), _dp(lon_out), _dp(lat_out), _dp(mask),
            _ip(i_in), _ip(j_in), _ip(i_out), _ip(j_out), _dp(area)]
    if order == 1:
        n = lib().create_xgrid_2dx2d_order1(*args)
        return n, i_in[:n].copy(), j_in[:n].copy(), i_out[:n].copy(), j_out[:n].copy(), area[:n].copy()
    clon, clat = np.empty(cap, dtype=np.float64), np.empty(cap, dtype=np.float64)
    n = lib().create_xgrid_2dx2d_order2(*args, _dp(clon), _dp(clat))
    return (n, i_in[:n].copy(), j_in[:n].copy(), i_out[:n].copy(), j_out[:n].copy(), area[:n].copy(),
            clon[:n].copy(), clat[:n].copy())


def create_xgrid_2dx2d_order1(nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in=None):
    """create_xgrid.c:621 -- returns (nxgrid, i_in, j_in, i_out, j_out, xgrid_area)."""
    return _create_xgrid(1, nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in)


def create_xgrid_2dx2d_order2(nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in=None):
    """create_xgrid.c:893 -- returns (nxgrid, i_in, j_in, i_out, j_out, xgrid_area, xgrid_clon, xgrid_clat)."""
    return _create_xgrid(2, nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in)


def conserve_interp(nx_src, ny_src, nx_dst, ny_dst, x_src, y_src, x_dst, y_dst, mask_src, data_src):
    """interp.c:262 -- first-order remap; returns data_dst[ny_dst*nx_dst]."""
    _lib.require_gpu()
    x_src, y_src, x_dst, y_dst = _f64(x_src), _f64(y_src), _f64(x_dst), _f64(y_dst)
    mask = _f64(np.ones(nx_src * ny_src) if mask_src is None else mask_src)
    data_src = _f64(data_src)
    out = np.empty(nx_dst * ny_dst, dtype=np.float64)
    lib().conserve_interp(nx_src, ny_src, nx_dst, ny_dst, _dp(x_src), _dp(y_src), _dp(x_dst), _dp(y_dst),
                          _dp(mask), _dp(data_src), _dp(out))
    return out


def create_xgrid_great_circle(nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in=None):
    """create_xgrid.c:1366 -- returns (nxgrid, i_in, j_in, i_out, j_out, xgrid_area, xgrid_clon, xgrid_clat); the
    centroid outputs are zero, as in the reference (:1446-1447)."""
    _lib.require_gpu()
    lon_in, lat_in, lon_out, lat_out = _f64(lon_in), _f64(lat_in), _f64(lon_out), _f64(lat_out)
    assert lon_in.size == (nlon_in + 1) * (nlat_in + 1) == lat_in.size
    assert lon_out.size == (nlon_out + 1) * (nlat_out + 1) == lat_out.size
    mask = _f64(np.ones(nlon_in * nlat_in) if mask_in is None else mask_in)
    cap = get_maxxgrid()
    i_in, j_in, i_out, j_out = (np.empty(cap, dtype=np.int32) for _ in range(4))
    area, clon, clat = (np.empty(cap, dtype=np.float64) for _ in range(3))
    n = lib().create_xgrid_great_circle(C.byref(C.c_int(nlon_in)), C.byref(C.c_int(nlat_in)), C.byref(C.c_int(nlon_out)),
                                        C.byref(C.c_int(nlat_out)), _dp(lon_in), _dp(lat_in), _dp(lon_out), _dp(lat_out),
                                        _dp(mask), _ip(i_in), _ip(j_in), _ip(i_out), _ip(j_out), _dp(area), _dp(clon), _dp(clat))
    return (n, i_in[:n].copy(), j_in[:n].copy(), i_out[:n].copy(), j_out[:n].copy(), area[:n].copy(),
            clon[:n].copy(), clat[:n].copy())


def get_grid_great_circle_area(nlon, nlat, lon, lat):
    """create_xgrid.c:98 -- area[nlat*nlon] (m^2), spherical excess of each cell."""
    _lib.require_gpu()
    lon, lat = _f64(lon), _f64(lat)
    area = np.empty(nlon * nlat, dtype=np.float64)
    lib().get_grid_great_circle_area(C.byref(C.c_int(nlon)), C.byref(C.c_int(nlat)), _dp(lon), _dp(lat), _dp(area))
    return area


def latlon2xyz(lon, lat):
    """mosaic_util.c:212 -- unit vectors (host libm, threaded)."""
    lon, lat = _f64(lon).reshape(-1), _f64(lat).reshape(-1)
    x, y, z = (np.empty(lon.size) for _ in range(3))
    lib().fg_latlon2xyz(lon.size, _dp(lon), _dp(lat), _dp(x), _dp(y), _dp(z))
    return x, y, z


def gc_clip_batch(a, b, device=0):
    """clip_2dx2d_great_circle on npairs quadrilateral pairs: a, b [npairs, 4, 3] unit vectors (clockwise).
    Returns (n_out [npairs], vertices [npairs, 16, 3], area [npairs])."""
    _lib.require_gpu()
    a, b = _f64(a), _f64(b)
    n = a.shape[0]
    out = np.zeros((n, 16, 3))
    n_out = np.zeros(n, dtype=np.int32)
    area = np.zeros(n)
    _lib.check(lib().fg_gc_clip_batch(n, _dp(a), _dp(b), _dp(out), _ip(n_out), _dp(area), device))
    return n_out, out, area


def conserve_interp_great_circle(nx_src, ny_src, nx_dst, ny_dst, x_src, y_src, x_dst, y_dst, mask_src, data_src):
    """interp.c:312 -- first-order remap through the great-circle exchange grid; returns data_dst[ny_dst*nx_dst]."""
    _lib.require_gpu()
    x_src, y_src, x_dst, y_dst = _f64(x_src), _f64(y_src), _f64(x_dst), _f64(y_dst)
    mask = _f64(np.ones(nx_src * ny_src) if mask_src is None else mask_src)
    data_src = _f64(data_src)
    out = np.empty(nx_dst * ny_dst, dtype=np.float64)
    lib().conserve_interp_great_circle(nx_src, ny_src, nx_dst, ny_dst, _dp(x_src), _dp(y_src), _dp(x_dst), _dp(y_dst),
                                       _dp(mask), _dp(data_src), _dp(out))
    return out


def create_xgrid_box(box_is_src, order, lon_b, lat_b, nxq, nyq, lon_q, lat_q, mask=None):
    """create_xgrid_1dx2d_order1/2 (box_is_src: the regular grid with 1-D bounds lon_b/lat_b is the source) or
    create_xgrid_2dx1d_order1/2 (it is the destination) -- create_xgrid.c:208-591.  mask is on the source cells.
    Returns (nxgrid, i_in, j_in, i_out, j_out, area[, clon, clat])."""
    _lib.require_gpu()
    lon_b, lat_b, lon_q, lat_q = _f64(lon_b).reshape(-1), _f64(lat_b).reshape(-1), _f64(lon_q), _f64(lat_q)
    nxb, nyb = lon_b.size - 1, lat_b.size - 1
    assert lon_q.size == (nxq + 1) * (nyq + 1) == lat_q.size
    nm = nxb * nyb if box_is_src else nxq * nyq
    mask = _f64(np.ones(nm) if mask is None else mask)
    assert mask.size == nm
    cap = get_maxxgrid()
    i_in, j_in, i_out, j_out = (np.empty(cap, dtype=np.int32) for _ in range(4))
    area, clon, clat = (np.empty(cap, dtype=np.float64) for _ in range(3))
    ci = lambda v: C.byref(C.c_int(v))
    if box_is_src:
        head = [ci(nxb), ci(nyb), ci(nxq), ci(nyq), _dp(lon_b), _dp(lat_b), _dp(lon_q), _dp(lat_q)]
        fn = lib().create_xgrid_1dx2d_order1 if order == 1 else lib().create_xgrid_1dx2d_order2
    else:
        head = [ci(nxq), ci(nyq), ci(nxb), ci(nyb), _dp(lon_q), _dp(lat_q), _dp(lon_b), _dp(lat_b)]
        fn = lib().create_xgrid_2dx1d_order1 if order == 1 else lib().create_xgrid_2dx1d_order2
    args = head + [_dp(mask), _ip(i_in), _ip(j_in), _ip(i_out), _ip(j_out), _dp(area)]
    if order == 2:
        args += [_dp(clon), _dp(clat)]
    n = fn(*args)
    out = (n, i_in[:n].copy(), j_in[:n].copy(), i_out[:n].copy(), j_out[:n].copy(), area[:n].copy())
    return out + ((clon[:n].copy(), clat[:n].copy()) if order == 2 else ())
