"""Order-2 input preparation: host mirror of the pieces of get_input_grid / get_input_data that feed
the second-order sweep -- tile contacts, halo update, calc_c2l_grid_info, grad_c2l, gradient mask
(tools/fregrid/fregrid_util.c:270-346, :2137-2216; tools/libfrencutils/gradient_c2l.c)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check

_dpt = C.POINTER(C.c_double)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(_dpt)


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def find_contacts(nx, ny, lonc, latc, max_contacts=64):
    """Line contacts between tiles, in read_mosaic_contact's convention (tools/libfrencutils/read_mosaic.c:655-777).
    Returns a dict of int32 arrays: tile1, tile2, istart1, iend1, jstart1, jend1, istart2, iend2, jstart2, jend2."""
    nt = len(nx)
    lon = [_f64(a).reshape(-1) for a in lonc]
    lat = [_f64(a).reshape(-1) for a in latc]
    keys = ["tile1", "tile2", "istart1", "iend1", "jstart1", "jend1", "istart2", "iend2", "jstart2", "jend2"]
    out = {k: np.zeros(max_contacts, dtype=np.int32) for k in keys}
    n = lib().fg_find_contacts(nt, (C.c_int * nt)(*nx), (C.c_int * nt)(*ny), (_dpt * nt)(*[_dp(a) for a in lon]),
                               (_dpt * nt)(*[_dp(a) for a in lat]), max_contacts, *[_ip(out[k]) for k in keys])
    if n < 0:
        raise ValueError(f"fg_find_contacts failed: {n}")
    return {k: v[:n].copy() for k, v in out.items()}


def c2l_grid_info(nx, ny, xt_halo, yt_halo, xc, yc):
    """calc_c2l_grid_info (gradient_c2l.c:368) on the host; returns a dict of arrays."""
    xt, yt, xc, yc = (_f64(a).reshape(-1) for a in (xt_halo, yt_halo, xc, yc))
    d = dict(dx=np.empty(nx * (ny + 1)), dy=np.empty((nx + 1) * ny), area=np.empty(nx * ny),
             edge_w=np.empty(ny + 1), edge_e=np.empty(ny + 1), edge_s=np.empty(nx + 1), edge_n=np.empty(nx + 1),
             en_n=np.empty(3 * nx * (ny + 1)), en_e=np.empty(3 * (nx + 1) * ny), vlon=np.empty(3 * nx * ny), vlat=np.empty(3 * nx * ny))
    rc = lib().fg_c2l_grid_info(nx, ny, _dp(xt), _dp(yt), _dp(xc), _dp(yc), *[_dp(d[k]) for k in
                                ("dx", "dy", "area", "edge_w", "edge_e", "edge_s", "edge_n", "en_n", "en_e", "vlon", "vlat")])
    if rc:
        raise ValueError(f"fg_c2l_grid_info failed: {rc}")
    return d


def halo_map(nx, ny, contacts):
    """Gather map of setup_boundary + update_halo (CENTER, halo 1): (map_off[ntiles+1], map[F])."""
    nt = len(nx)
    F = sum((a + 2) * (b + 2) for a, b in zip(nx, ny))
    off = np.zeros(nt + 1, dtype=np.int64)
    m = np.empty(F, dtype=np.int32)
    keys = ["tile1", "tile2", "istart1", "iend1", "jstart1", "jend1", "istart2", "iend2", "jstart2", "jend2"]
    c = {k: np.ascontiguousarray(contacts[k], dtype=np.int32) for k in keys}
    rc = lib().fg_halo_map(nt, (C.c_int * nt)(*nx), (C.c_int * nt)(*ny), len(c["tile1"]), *[_ip(c[k]) for k in keys],
                           off.ctypes.data_as(C.POINTER(C.c_long)), _ip(m))
    if rc:
        raise ValueError(f"fg_halo_map failed: {rc}")
    return off, m


class C2lPrep:
    """RAII wrapper of fg_c2l (include/fregrid_hip.h)."""

    def __init__(self, nx, ny, lonc, latc, lont, latt, contacts, device=0):
        _lib.require_gpu()
        nt = len(nx)
        keep = [[_f64(a).reshape(-1) for a in arrs] for arrs in (lonc, latc, lont, latt)]
        ptrs = [(_dpt * nt)(*[_dp(a) for a in arrs]) for arrs in keep]
        keys = ["tile1", "tile2", "istart1", "iend1", "jstart1", "jend1", "istart2", "iend2", "jstart2", "jend2"]
        c = {k: np.ascontiguousarray(contacts[k], dtype=np.int32) for k in keys}
        h = C.c_void_p()
        check(lib().fg_c2l_create(nt, (C.c_int * nt)(*nx), (C.c_int * nt)(*ny), *ptrs, len(c["tile1"]),
                                  *[_ip(c[k]) for k in keys], device, C.byref(h)))
        self._h = h
        self.ncells = int(lib().fg_c2l_ncells(h))
        self.F = int(lib().fg_c2l_halo_size(h))

    def destroy(self):
        if self._h is not None and self._h.value:
            lib().fg_c2l_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def centres(self):
        a, b = np.empty(self.F), np.empty(self.F)
        check(lib().fg_c2l_get_centres(self._h, _dp(a), _dp(b)))
        return a, b

    def set_stream(self, stream):
        check(lib().fg_c2l_set_stream(self._h, C.c_void_p(int(stream))))

    def sync(self):
        check(lib().fg_c2l_sync(self._h))

    def fill_halo(self, src_t, halo_t, nz):
        check(lib().fg_c2l_fill_halo(self._h, C.c_void_p(src_t.data_ptr() if src_t is not None else 0),
                                     C.c_void_p(halo_t.data_ptr()), nz))

    def gradient(self, halo_t, nz, grad_x_t, grad_y_t, grad_mask_t=None, has_missing=False, missing=0.0):
        check(lib().fg_c2l_gradient(self._h, C.c_void_p(halo_t.data_ptr()), nz, 1 if has_missing else 0, float(missing),
                                    C.c_void_p(grad_x_t.data_ptr()), C.c_void_p(grad_y_t.data_ptr()),
                                    C.c_void_p(grad_mask_t.data_ptr() if grad_mask_t is not None else 0)))

    @staticmethod
    def records_nb(nz):
        """levels per record for nz <= 8 levels: 2, 4 or 8"""
        return 8 if nz > 4 else (4 if nz > 2 else 2)

    def gradient_records(self, halo_t, nz, rec_t):
        """halo'd levels [nz, F] -> rec [ncells, 3, records_nb(nz)] = {field, grad_x, grad_y} per level: the layout
        XgridPlan.apply_records sweeps, with no level-major gradient arrays in between."""
        check(lib().fg_c2l_gradient_records(self._h, C.c_void_p(halo_t.data_ptr()), nz, C.c_void_p(rec_t.data_ptr())))

    def records(self, src_t, nz, rec_t):
        """unpadded levels [nz, ncells] -> rec [ncells, 3, records_nb(nz)] in one pass (fg_c2l_records): halo values are read
        from the neighbour tiles through the halo map, no halo'd copy is made."""
        check(lib().fg_c2l_records(self._h, C.c_void_p(src_t.data_ptr()), nz, C.c_void_p(rec_t.data_ptr())))
