"""fregrid's --remap_file without libnetcdf: thin wrappers over csrc/remap_file.c.

write_remap_file mirrors the WRITE branch of setup_conserve_interp (tools/fregrid/conserve_interp.c:368-445),
read_remap_file the READ branch's file access (:62-90 via tools/libfrencutils/read_mosaic.c:352-558)."""
import ctypes as C

import numpy as np

from ._lib import lib


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def write_remap_file(path, order, t_in, i_in, j_in, i_out, j_out, area, di_in=None, dj_in=None, isc=0, jsc=0):
    a = [np.ascontiguousarray(v, dtype=np.int32) for v in (t_in, i_in, j_in, i_out, j_out)]
    area = np.ascontiguousarray(area, dtype=np.float64)
    di = np.ascontiguousarray(di_in, dtype=np.float64) if di_in is not None else None
    dj = np.ascontiguousarray(dj_in, dtype=np.float64) if dj_in is not None else None
    rc = lib().fg_remap_write_interp(str(path).encode(), order, area.size, *[_ip(v) for v in a], _dp(area), _dp(di), _dp(dj), isc, jsc)
    if rc:
        raise IOError(lib().fg_remap_last_error().decode())


def read_remap_file(path, order):
    """Returns dict(t_in, i_in, j_in, i_out, j_out int32 0-based; area; di_in, dj_in for order 2)."""
    n = lib().fg_remap_read_size(str(path).encode())
    if n < 0:
        raise IOError(lib().fg_remap_last_error().decode())
    ints = {k: np.empty(n, dtype=np.int32) for k in ("t_in", "i_in", "j_in", "i_out", "j_out")}
    area = np.empty(n)
    di = np.empty(n) if order == 2 else None
    dj = np.empty(n) if order == 2 else None
    rc = lib().fg_remap_read(str(path).encode(), order, n, *[_ip(ints[k]) for k in ("t_in", "i_in", "j_in", "i_out", "j_out")],
                             _dp(area), _dp(di), _dp(dj))
    if rc:
        raise IOError(lib().fg_remap_last_error().decode())
    out = dict(ints)
    out["area"] = area
    if order == 2:
        out["di_in"], out["dj_in"] = di, dj
    return out
