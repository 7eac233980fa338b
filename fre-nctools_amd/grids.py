"""Synthetic grid generators (thin ctypes wrappers over csrc/grid_gen.c).

* gnomonic_ed_corners: make_hgrid --grid_type gnomonic_ed (tools/make_hgrid/create_gnomonic_cubic_grid.c:101)
  followed by fregrid's read-back of every second supergrid point (tools/fregrid/fregrid_util.c:227-232).
* latlon_corners: get_output_grid_by_size (tools/fregrid/fregrid_util.c:564-654).
Host-only C code: usable without a GPU.
"""
import ctypes as C

import numpy as np

from ._lib import lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def gnomonic_ed_corners(ni, shift_fac=18.0, via_degrees=True):
    """C<ni> cubed sphere: returns (lonc, latc), each [6, ni+1, ni+1] float64 radians."""
    lon = np.empty((6, ni + 1, ni + 1), dtype=np.float64)
    lat = np.empty_like(lon)
    rc = lib().fg_gnomonic_ed_corners(ni, float(shift_fac), 1 if via_degrees else 0, _dp(lon), _dp(lat))
    if rc:
        raise ValueError(f"fg_gnomonic_ed_corners({ni}) failed: {rc}")
    return lon, lat


def latlon_corners(nlon, nlat, lonbegin=0.0, lonend=360.0, latbegin=-90.0, latend=90.0, center_y=True):
    """Regular lat-lon target grid: returns (lonc, latc), each [nlat+1, nlon+1] float64 radians."""
    lon = np.empty((nlat + 1, nlon + 1), dtype=np.float64)
    lat = np.empty_like(lon)
    rc = lib().fg_latlon_corners(nlon, nlat, lonbegin, lonend, latbegin, latend, 1 if center_y else 0, _dp(lon), _dp(lat))
    if rc:
        raise ValueError(f"fg_latlon_corners failed: {rc}")
    return lon, lat


def gnomonic_ed_grid(ni, shift_fac=18.0, via_degrees=True):
    """C<ni> cubed sphere corners and T-cell centres: (lonc, latc [6, ni+1, ni+1], lont, latt [6, ni, ni])."""
    lon = np.empty((6, ni + 1, ni + 1), dtype=np.float64)
    lat = np.empty_like(lon)
    lont = np.empty((6, ni, ni), dtype=np.float64)
    latt = np.empty_like(lont)
    rc = lib().fg_gnomonic_ed_grid(ni, float(shift_fac), 1 if via_degrees else 0, _dp(lon), _dp(lat), _dp(lont), _dp(latt))
    if rc:
        raise ValueError(f"fg_gnomonic_ed_grid({ni}) failed: {rc}")
    return lon, lat, lont, latt


def tripolar_corners(nlon, nlat, xbnd=(-280.0, 80.0), ybnd=(-82.0, 90.0), lat_join=65.0):
    """Tripolar ocean grid corners (bounds as tests/fregrid/latlon:36-37): (lonc, latc) [nlat+1, nlon+1] radians."""
    lon = np.empty((nlat + 1, nlon + 1), dtype=np.float64)
    lat = np.empty_like(lon)
    rc = lib().fg_tripolar_corners(nlon, nlat, xbnd[0], xbnd[1], ybnd[0], ybnd[1], lat_join, _dp(lon), _dp(lat))
    if rc:
        raise ValueError(f"fg_tripolar_corners failed: {rc}")
    return lon, lat
