"""csrc/field_file.c (fg_nc_*): the classic-netCDF reader / writer behind the field and grid files, checked in both
directions against an independent implementation of the format (scipy.io.netcdf_file, CDF-1 / CDF-2), plus CDF-5 and
record-variable round trips.  Mirrors what mpp_io.c does for fregrid (mpp_get_var_value_block / mpp_put_var_value_block,
attributes): tools/libfrencutils/mpp_io.c:443-481,487-,1349-."""
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

from conftest import load_package

fg = load_package()
NC = fg.field_io


def _field(nt, nz, ny, nx, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((nt, nz, ny, nx))


@pytest.mark.parametrize("version", [1, 2])
def test_our_file_read_by_scipy(tmp_path, version):
    p = str(tmp_path / "w.nc")
    f = NC.NcFile.create(p, version)
    dt, dz, dy, dx = f.def_dim("time", 0), f.def_dim("pfull", 3), f.def_dim("lat", 5), f.def_dim("lon", 7)
    vt = f.def_var("time", NC.NC_DOUBLE, [dt])
    vtemp = f.def_var("temp", NC.NC_FLOAT, [dt, dz, dy, dx])
    vps = f.def_var("ps", NC.NC_SHORT, [dt, dy, dx])
    vlat = f.def_var("lat", NC.NC_DOUBLE, [dy])
    f.put_att(vtemp, "units", "K"); f.put_att(vtemp, "missing_value", -1.0e10, NC.NC_FLOAT)
    f.put_att(vps, "scale_factor", 0.5); f.put_att(vps, "add_offset", 1000.0)
    f.put_att(-1, "history", "written by fg_nc")
    f.enddef()
    temp = _field(4, 3, 5, 7, 1).astype(np.float32)
    ps = (np.arange(4 * 5 * 7).reshape(4, 5, 7) % 30000).astype(np.int16)
    f.put_vara("lat", np.linspace(-80, 80, 5))
    for t in range(4):                                   # record by record, as write_field_data does
        f.put_vara("time", np.array([float(t)]), [t], [1])
        f.put_vara("temp", temp[t:t + 1], [t, 0, 0, 0], [1, 3, 5, 7])
        f.put_vara("ps", ps[t:t + 1], [t, 0, 0], [1, 5, 7])
    f.close()
    with netcdf_file(p, "r", mmap=False) as s:
        assert s.version_byte == version
        assert s.dimensions["time"] is None and s.dimensions["lon"] == 7
        assert np.array_equal(s.variables["temp"][:], temp) and s.variables["temp"].units == b"K"
        assert np.array_equal(s.variables["ps"][:], ps) and s.variables["ps"].scale_factor == 0.5
        assert np.array_equal(s.variables["time"][:], np.arange(4.0))
        assert np.array_equal(s.variables["lat"][:], np.linspace(-80, 80, 5))
        assert s.history == b"written by fg_nc"
        assert np.float32(s.variables["temp"].missing_value) == np.float32(-1.0e10)


@pytest.mark.parametrize("version", [1, 2])
def test_scipy_file_read_by_us(tmp_path, version):
    p = str(tmp_path / "r.nc")
    temp = _field(3, 4, 6, 9, 2).astype(np.float32)
    sst = _field(3, 1, 6, 9, 3)[:, 0]
    orog = (np.arange(54).reshape(6, 9) * 3).astype(np.int32)
    with netcdf_file(p, "w", version=version) as s:
        s.createDimension("time", None); s.createDimension("z", 4); s.createDimension("y", 6); s.createDimension("x", 9)
        v = s.createVariable("temp", "f", ("time", "z", "y", "x")); v[:] = temp; v.missing_value = np.float32(1.0e20); v.long_name = "temperature"
        v = s.createVariable("sst", "d", ("time", "y", "x")); v[:] = sst; v.scale_factor = 2.0; v.add_offset = -3.0
        v = s.createVariable("orog", "i", ("y", "x")); v[:] = orog
        v = s.createVariable("time", "d", ("time",)); v[:] = [0.5, 1.5, 2.5]
    f = NC.NcFile(p)
    assert f.dims() == {"time": 3, "z": 4, "y": 6, "x": 9} and f.numrecs == 3
    assert set(f.variables()) == {"temp", "sst", "orog", "time"}
    assert f.inq_var("temp")["type"] == NC.NC_FLOAT and f.inq_var("temp")["shape"] == (3, 4, 6, 9)
    assert np.array_equal(f.get_vara("temp"), temp)
    # hyperslabs: one time level, a z range, partial rows and columns
    assert np.array_equal(f.get_vara("temp", [1, 1, 0, 0], [1, 2, 6, 9]), temp[1:2, 1:3])
    assert np.array_equal(f.get_vara("temp", [2, 0, 2, 3], [1, 4, 3, 5]), temp[2:3, :, 2:5, 3:8])
    assert np.array_equal(f.get_vara("sst", [0, 5, 0], [3, 1, 9]), sst[:, 5:6])
    assert np.array_equal(f.get_vara("orog"), orog)
    # widening as nc_get_vara_double (what NC_FLOAT goes through in get_input_data)
    d = f.get_vara("temp", [1, 0, 0, 0], [1, 4, 6, 9], as_double=True)
    assert d.dtype == np.float64 and np.array_equal(d, temp[1:2].astype(np.float64))
    assert np.array_equal(f.get_vara("orog", as_double=True), orog.astype(np.float64))
    assert f.get_att("sst", "scale_factor") == 2.0 and f.get_att("sst", "add_offset") == -3.0
    assert f.get_att("temp", "long_name") == "temperature" and f.get_att("temp", "nope") is None
    assert np.float32(f.get_att("temp", "missing_value")) == np.float32(1.0e20)
    with pytest.raises(IOError):
        f.get_vara("temp", [0, 0, 0, 0], [4, 4, 6, 9])      # beyond numrecs
    f.close()


def test_cdf5_and_double_conversion_round_trip(tmp_path):
    p = str(tmp_path / "c5.nc")
    a = _field(2, 3, 4, 5, 5)
    f = NC.NcFile.create(p, 5)
    dt, dz, dy, dx = f.def_dim("t", 0), f.def_dim("z", 3), f.def_dim("y", 4), f.def_dim("x", 5)
    f.def_var("a", NC.NC_DOUBLE, [dt, dz, dy, dx]); f.def_var("b", NC.NC_FLOAT, [dt, dz, dy, dx]); f.def_var("c", NC.NC_SHORT, [dy, dx])
    f.enddef()
    f.put_vara("a", a); f.put_vara("b", a)                  # double -> float through fg_nc_put_vara_double: the C cast
    f.put_vara("c", np.arange(20.0).reshape(4, 5) - 7.6)   # double -> short: (short) cast, as write_field_data
    f.close()
    assert open(p, "rb").read(4) == b"CDF\x05"
    f = NC.NcFile(p)
    assert np.array_equal(f.get_vara("a"), a)
    assert np.array_equal(f.get_vara("b"), a.astype(np.float32))
    assert np.array_equal(f.get_vara("c"), (np.arange(20.0).reshape(4, 5) - 7.6).astype(np.int16))
    assert np.array_equal(f.get_vara("b", [1, 2, 0, 0], [1, 1, 4, 5], as_double=True), a.astype(np.float32).astype(np.float64)[1:2, 2:3])
    f.close()


def test_read_field_levels_follows_get_input_data(tmp_path):
    """One (t, z-range, y, x) hyperslab per tile file, tiles back to back, in the file type + scale / offset / missing."""
    files = []
    ref = []
    for tile in range(2):
        p = str(tmp_path / f"in.tile{tile + 1}.nc")
        a = _field(3, 5, 4, 6, 10 + tile).astype(np.float32)
        with netcdf_file(p, "w", version=2) as s:
            s.createDimension("time", None); s.createDimension("z", 5); s.createDimension("y", 4); s.createDimension("x", 6)
            v = s.createVariable("u", "f", ("time", "z", "y", "x")); v[:] = a; v.missing_value = np.float32(-999.0); v.scale_factor = 0.25
        files.append(NC.NcFile(p)); ref.append(a)
    data, meta = fg.read_field_levels(files, "u", level_t=2, kstart=1, nz=3)
    assert data.dtype == np.float32 and data.shape == (3, 2 * 24)
    assert np.array_equal(data[:, :24], ref[0][2, 1:4].reshape(3, -1)) and np.array_equal(data[:, 24:], ref[1][2, 1:4].reshape(3, -1))
    assert meta["type"] == NC.NC_FLOAT and meta["scale"] == 0.25 and meta["offset"] == 0.0 and meta["missing"] == -999.0
    for f in files:
        f.close()


def test_rejects_hdf5_and_garbage(tmp_path):
    p = str(tmp_path / "h5.nc")
    open(p, "wb").write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(IOError, match="netCDF-4"):
        NC.NcFile(p)
    open(p, "wb").write(b"not a file")
    with pytest.raises(IOError):
        NC.NcFile(p)


def test_malformed_cdf5_headers_are_refused_under_asan(tmp_path):
    """a CDF-5 header is 64-bit -- an attribute count whose byte size wraps (2^61 doubles = 0 bytes mod 2^64) must be refused
    before it is multiplied, not copied; and a header that fails to parse must not make the reader re-read ever larger pieces of
    a file that has no more bytes.  csrc/field_file.c compiled with -fsanitize=address,undefined and run on crafted files."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "fre-nctools_amd", "csrc", "field_file.c")
    drv = tmp_path / "drv.c"
    drv.write_text('#include <stdio.h>\n#include "fregrid_hip.h"\n'
                   'int main(int c, char **v) { int k, bad = 0; for (k = 1; k < c; k++) { fg_ncfile *f = 0; int rc = fg_nc_open(v[k], &f);\n'
                   '  printf("%s -> %d %s\\n", v[k], rc, rc ? fg_nc_last_error() : "ok"); if (!rc) { fg_nc_close(f); bad = 1; } } return bad; }\n')
    exe = str(tmp_path / "drv")
    r = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(root, "include"),
                        str(drv), src, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    u32 = lambda v: struct.pack(">I", v)
    u64 = lambda v: struct.pack(">Q", v)
    name = lambda s: u64(len(s)) + s + b"\0" * (-len(s) % 4)
    head = b"CDF\x05" + u64(0) + u32(0) + u64(0)                       # numrecs 0, no dimensions
    cases = {
        "wrap.nc": head + u32(12) + u64(1) + name(b"a") + u32(6) + u64((1 << 61) + 1) + b"\0" * 64,        # global attribute: 2^61 + 1 doubles
        "huge.nc": head + u32(12) + u64(1) + name(b"a") + u32(6) + u64(1 << 40) + b"\0" * 64,              # 8 TiB of attribute values
        "trunc.nc": (head + u32(12) + u64(3) + name(b"a") + u32(6) + u64(2) + b"\0" * 16)[:-9],            # cut in the middle of a value
        "badtype.nc": head + u32(12) + u64(1) + name(b"a") + u32(99) + u64(1) + b"\0" * 16,                # unknown attribute type
    }
    paths = []
    for fn, blob in cases.items():
        p = tmp_path / fn
        p.write_bytes(blob); paths.append(str(p))
    r = subprocess.run([exe] + paths, capture_output=True, text=True, timeout=60, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert r.stdout.count("malformed header") == len(cases), r.stdout
    # ... and the unsigned / 64-bit attribute types of CDF-5 decode to their values
    good = tmp_path / "u.nc"
    good.write_bytes(head + u32(12) + u64(2) + name(b"u") + u32(9) + u64(1) + struct.pack(">I", 4000000000) +
                     name(b"q") + u32(10) + u64(1) + struct.pack(">q", -5) + u32(0) + u64(0))
    f = NC.NcFile(str(good))
    import ctypes as C
    val = (C.c_double * 1)()
    assert fg.lib().fg_nc_get_att_double(f._h, -1, b"u", val, 1) == 1 and val[0] == 4000000000.0
    assert fg.lib().fg_nc_get_att_double(f._h, -1, b"q", val, 1) == 1 and val[0] == -5.0
    f.close()
