"""The oracle against the known answers of the reference's own GPU unit tests (t_gpu/, see tests/test_gpu_reference_kats.py
for the citations): analytic cell records of a 36 x 4 lat-lon grid, and identical 359 x 2 grids giving the identity
exchange grid.  CPU only; pins the checker itself to the reference's KATs."""
import numpy as np

import orc

D2R = np.pi / 180.0


def latlon_grid(nlon, nlat, lon0, dlon, lat0, dlat):
    lon = (lon0 + dlon * np.arange(nlon + 1)) * D2R
    lat = (lat0 + dlat * np.arange(nlat + 1)) * D2R
    return np.ascontiguousarray(np.tile(lon, (nlat + 1, 1))), np.ascontiguousarray(np.repeat(lat[:, None], nlon + 1, axis=1))


def test_oracle_cell_struct_kat_36x4():
    nlon, nlat, dlon, dlat = 36, 4, 10.0, 30.0
    lo, la = latlon_grid(nlon, nlat, 0.0, dlon, -30.0, dlat)
    o = orc.orc_cell_struct(nlon, nlat, lo, la)
    jj, ii = np.meshgrid(np.arange(nlat), np.arange(nlon), indexing="ij")
    lat_min = ((-30.0 + dlat * jj) * D2R).ravel(); lat_max = ((-30.0 + dlat * (jj + 1)) * D2R).ravel()
    lon_min = (dlon * ii * D2R).ravel(); lon_max = (dlon * (ii + 1) * D2R).ravel()
    sel = (jj < nlat - 1).ravel()              # the row ending at the pole gets fix_lon's pole treatment in the legacy code
    assert np.all(o["nvert"][sel] == 4)
    assert np.max(np.abs(o["lat_min"] - lat_min)) < 1e-7 and np.max(np.abs(o["lat_max"] - lat_max)) < 1e-7
    assert np.max(np.abs(o["lon_min"][sel] - lon_min[sel])) < 1e-7 and np.max(np.abs(o["lon_max"][sel] - lon_max[sel])) < 1e-7
    assert np.max(np.abs(o["lon_avg"][sel] - 0.5 * (lon_min + lon_max)[sel])) < 1e-7
    vlon = o["vlon"].reshape(nlon * nlat, -1); vlat = o["vlat"].reshape(nlon * nlat, -1)
    assert np.max(np.abs(vlon[sel, :4] - np.stack([lon_min, lon_max, lon_max, lon_min], axis=1)[sel])) < 1e-7
    assert np.max(np.abs(vlat[sel, :4] - np.stack([lat_min, lat_min, lat_max, lat_max], axis=1)[sel])) < 1e-7


def test_oracle_identical_grids_give_identity_exchange_grid():
    nlon, nlat = 359, 2
    lo, la = latlon_grid(nlon, nlat, 0.0, 1.0, 0.0, 30.0)
    ncells = nlon * nlat
    for order in (1, 2):
        o = orc.orc_create_xgrid(order, nlon, nlat, nlon, nlat, lo, la, lo, la)
        assert o["n"] == ncells
        assert np.array_equal(o["j_in"].astype(np.int64) * nlon + o["i_in"], np.arange(ncells))
        assert np.array_equal(o["j_out"].astype(np.int64) * nlon + o["i_out"], np.arange(ncells))
    if orc.ref_available():                    # the compiled reference itself gives the same identity list
        r = orc.ref_create_xgrid(1, nlon, nlat, nlon, nlat, lo, la, lo, la)
        assert r["n"] == ncells
        assert np.array_equal(r["j_in"].astype(np.int64) * nlon + r["i_in"], np.arange(ncells))
        assert np.array_equal(r["j_out"].astype(np.int64) * nlon + r["i_out"], np.arange(ncells))
