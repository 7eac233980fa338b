"""The known-answer cases of the reference's own GPU unit tests (t_gpu/), run through this library.

* t_gpu/test_get_grid_cell_struct/test_get_grid_cell_struct.c:34-40,103-143: 36 x 4 lat-lon grid (10 x 30 degree cells,
  latitudes -30..90), analytic per-cell lon/lat minima, maxima, mean longitude and the four vertices in the order
  (min,min) (max,min) (max,max) (min,max), tolerance 1e-7.
* t_gpu/test_get_upbound_nxcells_2dx2d/test_get_upbound_nxcells_2dx2d.c:91-134: identical 359 x 2 grids (1 x 30 degree
  cells): exactly one bounding-box candidate per source cell, and it is the cell itself.
* t_gpu/test_get_interp_order1/test_get_interp_order1.c:65-107: the same grids through create_xgrid order 1: as many
  exchange cells as cells, parent indices = identity.
The expected values are the analytic ones those tests construct (data, restated here), not copied code."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
D2R = np.pi / 180.0
TOL = 1.e-7                                   # test_get_grid_cell_struct.c:40


def latlon_grid(nlon, nlat, lon0, dlon, lat0, dlat):
    lon = (lon0 + dlon * np.arange(nlon + 1)) * D2R
    lat = (lat0 + dlat * np.arange(nlat + 1)) * D2R
    return np.ascontiguousarray(np.tile(lon, (nlat + 1, 1))), np.ascontiguousarray(np.repeat(lat[:, None], nlon + 1, axis=1))


def test_kat_get_grid_cell_struct_36x4(fg, gpu_ok):
    nlon, nlat, dlon, dlat = 36, 4, 10.0, 30.0
    lo, la = latlon_grid(nlon, nlat, 0.0, dlon, -30.0, dlat)
    g = fg.GridConfig(nlon, nlat, lo, la)
    plan = fg.XgridPlan.create(1, [g], g)
    d = plan.get_cell_struct(1, nlon * nlat)
    o = orc.orc_cell_struct(nlon, nlat, lo, la)
    plan.destroy()
    # the whole grid, pole row included, against the oracle (bit for bit, as in test_cell_struct_bitwise)
    for k in ("lat_min", "lat_max", "lon_min", "lon_max", "lon_avg", "vlon", "vlat"):
        assert np.array_equal(d[k].view(np.uint64), o[k].view(np.uint64)), k
    assert np.array_equal(d["nvert"], o["nvert"])
    # the reference test's analytic answers.  Its top row ends at 90 degrees: there the legacy get_grid_cell_struct
    # (create_xgrid.c:991-1016) applies fix_lon's pole treatment, so the four-vertex answers are asserted for the three
    # rows below it and the pole row is covered by the oracle comparison above.
    jj, ii = np.meshgrid(np.arange(nlat), np.arange(nlon), indexing="ij")
    lat_min = ((-30.0 + dlat * jj) * D2R).ravel(); lat_max = ((-30.0 + dlat * (jj + 1)) * D2R).ravel()
    lon_min = (dlon * ii * D2R).ravel(); lon_max = (dlon * (ii + 1) * D2R).ravel()
    lon_cent = ((dlon * ii + 0.5 * dlon) * D2R).ravel()
    sel = (jj < nlat - 1).ravel()
    assert np.all(d["nvert"][sel] == 4)
    for got, want in ((d["lat_min"], lat_min), (d["lat_max"], lat_max), (d["lon_min"], lon_min), (d["lon_max"], lon_max),
                      (d["lon_avg"], lon_cent)):
        assert np.max(np.abs(got[sel] - want[sel])) < TOL
    vlon = d["vlon"].reshape(nlon * nlat, -1); vlat = d["vlat"].reshape(nlon * nlat, -1)
    want_lon = np.stack([lon_min, lon_max, lon_max, lon_min], axis=1)
    want_lat = np.stack([lat_min, lat_min, lat_max, lat_max], axis=1)
    assert np.max(np.abs(vlon[sel, :4] - want_lon[sel])) < TOL
    assert np.max(np.abs(vlat[sel, :4] - want_lat[sel])) < TOL
    # min/max/latitudes of the pole row are analytic too
    assert np.max(np.abs(d["lat_min"] - lat_min)) < TOL and np.max(np.abs(d["lat_max"] - lat_max)) < TOL


def test_kat_identical_grids_upbound_and_order1(fg, gpu_ok):
    nlon, nlat = 359, 2
    lo, la = latlon_grid(nlon, nlat, 0.0, 1.0, 0.0, 30.0)
    ncells = nlon * nlat
    g = fg.GridConfig(nlon, nlat, lo, la)
    plan = fg.XgridPlan.create(1, [g], g)
    st = plan.stats()
    x = plan.get_xgrid()
    plan.destroy()
    # get_upbound_nxcells_2dx2d: one candidate per source cell (ij2_start = ij2_end = ij1), upbound = ncells
    assert st["pairs"] == ncells
    # create_xgrid order 1: nxcells = ncells, parents = identity, in source-cell order
    assert len(x["area"]) == ncells
    ij_in = x["j_in"].astype(np.int64) * nlon + x["i_in"]
    ij_out = x["j_out"].astype(np.int64) * nlon + x["i_out"]
    assert np.array_equal(ij_in, np.arange(ncells)) and np.array_equal(ij_out, np.arange(ncells))
    # and the exchange areas are the cell areas
    area = fg.get_grid_area(nlon, nlat, lo, la)
    assert np.max(np.abs(x["area"] / area.ravel() - 1)) < 1e-12
    # the B1 mirror gives the same list
    n, i_in, j_in, i_out, j_out, xarea = fg.create_xgrid_2dx2d_order1(nlon, nlat, nlon, nlat, lo, la, lo, la)
    assert n == ncells
    assert np.array_equal(j_in.astype(np.int64) * nlon + i_in, np.arange(ncells))
    assert np.array_equal(j_out.astype(np.int64) * nlon + i_out, np.arange(ncells))
    assert np.array_equal(xarea.view(np.uint64), x["area"].view(np.uint64))
