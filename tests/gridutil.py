"""Shared helpers for building test inputs (cubed-sphere source tiles, lat-lon targets)."""
import ctypes as C

import numpy as np

import orc

D2R = np.pi / 180
R2D = 180 / np.pi


def ref_gnomonic_corners(ni, centres=False, supergrid=False):
    """C<ni> corners from the REFERENCE generator (create_gnomonic_cubic_grid.c:101 compiled in oracle/_ref),
    with fregrid's read-back (every 2nd supergrid point, degrees * D2R; fregrid_util.c:227-232).
    The generator chats on stderr; that is the reference's behaviour."""
    L = orc.ref()
    nx = 2 * ni
    nxp = nx + 1
    nlon = (C.c_int * 6)(*([nx] * 6))
    nlat = (C.c_int * 6)(*([nx] * 6))
    x = np.zeros(6 * nxp * nxp)
    y = np.zeros(6 * nxp * nxp)
    dx = np.zeros(6 * nx * nxp)
    dy = np.zeros(6 * nxp * nx)
    area = np.zeros(6 * nx * nx)
    adx = np.zeros(6 * nxp * nxp)
    ady = np.zeros(6 * nxp * nxp)
    nest = (C.c_int * 128)()
    dp = C.POINTER(C.c_double)
    f = L.create_gnomonic_cubic_grid
    f.restype = None
    f.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)] + [dp] * 7 + \
                 [C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int] + \
                 [C.POINTER(C.c_int)] * 6 + [C.c_int, C.c_int]
    P = lambda a: a.ctypes.data_as(dp)
    f(b"gnomonic_ed", nlon, nlat, P(x), P(y), P(dx), P(dy), P(area), P(adx), P(ady),
      18.0, 0, 0, 1.0, 0.0, 0.0, 0, nest, nest, nest, nest, nest, nest, 0, 0)
    x = x.reshape(6, nxp, nxp)
    y = y.reshape(6, nxp, nxp)
    if supergrid:
        return x, y                                  # degrees, as make_hgrid writes them
    if centres:                                      # T-cell centres: odd supergrid points (fregrid_util.c:238-243)
        return (np.ascontiguousarray(x[:, ::2, ::2] * D2R), np.ascontiguousarray(y[:, ::2, ::2] * D2R),
                np.ascontiguousarray(x[:, 1::2, 1::2] * D2R), np.ascontiguousarray(y[:, 1::2, 1::2] * D2R))
    return np.ascontiguousarray(x[:, ::2, ::2] * D2R), np.ascontiguousarray(y[:, ::2, ::2] * D2R)


def analytic_field(lon_c, lat_c):
    """10*sin(lon+lat): the field of tests/create_daily_tile_files.c:144, evaluated at cell centres."""
    return 10.0 * np.sin(lon_c + lat_c)


def cell_centres(lonc, latc):
    """Crude cell centres (mean of the 4 corners in 3-D) -- only used to synthesise smooth fields."""
    x = np.cos(latc) * np.cos(lonc)
    y = np.cos(latc) * np.sin(lonc)
    z = np.sin(latc)
    avg = lambda a: 0.25 * (a[:-1, :-1] + a[1:, :-1] + a[:-1, 1:] + a[1:, 1:])
    xm, ym, zm = avg(x), avg(y), avg(z)
    r = np.sqrt(xm * xm + ym * ym + zm * zm)
    lon = np.arctan2(ym, xm)
    lon = np.where(lon < 0, lon + 2 * np.pi, lon)
    return lon, np.arcsin(zm / r)
