"""The remap file without libnetcdf (csrc/remap_file.c): checked against an independent classic-netCDF
implementation (scipy.io.netcdf_file) in both directions, for the schema of tools/fregrid/conserve_interp.c:382-439 and the
conversions of tools/libfrencutils/read_mosaic.c:429-435,544-552.  The contract of the reference's own reader test
(t_gpu/test_read_remap_file/test_make_remap_file_conserve.py:53-151: 1-based (i,j) pairs flattened (ncells,2), tile1
per cell, area and distance order kept) is what these files encode."""
import numpy as np
import pytest
from scipy.io import netcdf_file

GAREA = 4 * np.pi * 6371000.0 ** 2


def _sample(n, order, seed=0):
    rng = np.random.default_rng(seed)
    d = dict(t_in=rng.integers(0, 6, n).astype(np.int32), i_in=rng.integers(0, 96, n).astype(np.int32),
             j_in=rng.integers(0, 96, n).astype(np.int32), i_out=rng.integers(0, 360, n).astype(np.int32),
             j_out=rng.integers(0, 180, n).astype(np.int32), area=rng.uniform(1e6, 1e10, n))
    if order == 2:
        d["di_in"] = rng.normal(0, 1e-3, n)
        d["dj_in"] = rng.normal(0, 1e-3, n)
    return d


@pytest.mark.parametrize("order,n", [(1, 1000), (2, 777), (2, 0), (1, 1)])
def test_written_file_is_valid_netcdf_with_the_reference_schema(fg, tmp_path, order, n):
    d = _sample(n, order)
    path = tmp_path / "remap.tile1.nc"
    fg.write_remap_file(path, order, d["t_in"], d["i_in"], d["j_in"], d["i_out"], d["j_out"], d["area"], d.get("di_in"), d.get("dj_in"))
    if n == 0:
        assert fg.lib().fg_remap_read_size(str(path).encode()) == 0
        return
    with netcdf_file(str(path), "r", mmap=False) as f:
        assert f.version_byte == 2
        assert f.dimensions["string"] == 255 and f.dimensions["ncells"] == n and f.dimensions["two"] == 2
        assert f.variables["tile1"].standard_name == b"tile_number_in_mosaic1"
        assert f.variables["tile1_cell"].standard_name == b"parent_cell_indices_in_mosaic1"
        assert f.variables["tile2_cell"].standard_name == b"parent_cell_indices_in_mosaic2"
        assert f.variables["xgrid_area"].standard_name == b"exchange_grid_area" and f.variables["xgrid_area"].units == b"m2"
        assert np.array_equal(f.variables["tile1"][:], d["t_in"] + 1)
        assert np.array_equal(f.variables["tile1_cell"][:], np.stack([d["i_in"] + 1, d["j_in"] + 1], axis=1))
        assert np.array_equal(f.variables["tile2_cell"][:], np.stack([d["i_out"] + 1, d["j_out"] + 1], axis=1))
        assert np.array_equal(f.variables["xgrid_area"][:], d["area"])
        if order == 2:
            assert f.variables["tile1_distance"].standard_name == b"distance_from_parent1_cell_centroid"
            assert np.array_equal(f.variables["tile1_distance"][:], np.stack([d["di_in"], d["dj_in"]], axis=1))
        else:
            assert "tile1_distance" not in f.variables
    r = fg.read_remap_file(path, order)
    for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
        assert np.array_equal(r[k], d[k])
    assert np.array_equal(r["area"], d["area"] / GAREA * GAREA)          # read_mosaic.c:432 then conserve_interp.c:86
    if order == 2:
        assert np.array_equal(r["di_in"], d["di_in"]) and np.array_equal(r["dj_in"], d["dj_in"])


@pytest.mark.parametrize("version", [1, 2])
def test_reads_files_written_by_another_netcdf_implementation(fg, tmp_path, version):
    n, order = 500, 2
    d = _sample(n, order, seed=3)
    path = tmp_path / f"other_v{version}.nc"
    with netcdf_file(str(path), "w", version=version) as f:
        f.history = "written by scipy"                                 # a global attribute the reader must skip
        f.createDimension("two", 2)
        f.createDimension("ncells", n)                                 # different dimension/variable order than ours
        v = f.createVariable("xgrid_area", "d", ("ncells",)); v[:] = d["area"]; v.units = "m2"
        v = f.createVariable("tile1_distance", "d", ("ncells", "two")); v[:] = np.stack([d["di_in"], d["dj_in"]], axis=1)
        v = f.createVariable("tile2_cell", "i", ("ncells", "two")); v[:] = np.stack([d["i_out"] + 1, d["j_out"] + 1], axis=1)
        v = f.createVariable("tile1_cell", "i", ("ncells", "two")); v[:] = np.stack([d["i_in"] + 1, d["j_in"] + 1], axis=1)
        v = f.createVariable("tile1", "i", ("ncells",)); v[:] = d["t_in"] + 1
    r = fg.read_remap_file(path, order)
    for k in ("t_in", "i_in", "j_in", "i_out", "j_out", "di_in", "dj_in"):
        assert np.array_equal(r[k], d[k]), k
    assert np.array_equal(r["area"], d["area"] / GAREA * GAREA)
    r1 = fg.read_remap_file(path, 1)                                   # order-1 read of an order-2 file ignores the distances
    assert "di_in" not in r1 and np.array_equal(r1["i_in"], d["i_in"])


def test_reader_errors(fg, tmp_path):
    p = tmp_path / "hdf5.nc"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(IOError, match="netCDF-4"):
        fg.read_remap_file(p, 1)
    p2 = tmp_path / "junk.nc"
    p2.write_bytes(b"not a netcdf file at all")
    with pytest.raises(IOError):
        fg.read_remap_file(p2, 1)
    with pytest.raises(IOError):
        fg.read_remap_file(tmp_path / "missing.nc", 1)
    with netcdf_file(str(tmp_path / "nofield.nc"), "w", version=2) as f:
        f.createDimension("ncells", 3)
        f.createVariable("tile1", "i", ("ncells",))[:] = [1, 1, 1]
    with pytest.raises(IOError, match="tile1_cell"):
        fg.read_remap_file(tmp_path / "nofield.nc", 1)
