"""Field / grid files and the streamed sweep (include/fregrid_hip.h: fg_nc_*, fg_sweep_*).

``NcFile`` binds the classic-netCDF reader / writer of csrc/field_file.c -- the operations fregrid performs on its files
through tools/libfrencutils/mpp_io.c.  ``read_field_levels`` follows get_input_data (tools/fregrid/fregrid_util.c:2036-2165)
up to the point where the data leave the host: the hyperslab is returned in the FILE type with the variable's
scale_factor / add_offset / missing_value, so that ``Sweep`` can move the narrow type across PCIe and widen on the device.
"""
import ctypes as C

import numpy as np

from ._lib import lib, check

NC_BYTE, NC_CHAR, NC_SHORT, NC_INT, NC_FLOAT, NC_DOUBLE = 1, 2, 3, 4, 5, 6
_NP = {NC_BYTE: np.int8, NC_CHAR: np.uint8, NC_SHORT: np.int16, NC_INT: np.int32, NC_FLOAT: np.float32, NC_DOUBLE: np.float64}
_NC = {np.dtype(v): k for k, v in _NP.items() if k != NC_CHAR}
FG_ERR_NOTFOUND = -10


def nc_type_of(dtype):
    return _NC[np.dtype(dtype)]


def _nc_check(rc):
    if rc < 0:
        raise IOError(lib().fg_nc_last_error().decode())
    return rc


class NcFile:
    """A classic netCDF file (CDF-1 / 2 / 5).  ``NcFile(path)`` opens for reading, ``NcFile.create(path, version)`` for writing."""

    def __init__(self, path, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            _nc_check(lib().fg_nc_open(str(path).encode(), C.byref(self._h)))

    @classmethod
    def create(cls, path, version=2):
        h = C.c_void_p()
        _nc_check(lib().fg_nc_create(str(path).encode(), int(version), C.byref(h)))
        return cls(path, _handle=h)

    # ---- define mode (mpp_def_dim, mpp_def_var, mpp_def_*_att, mpp_end_def)
    def def_dim(self, name, length):
        return _nc_check(lib().fg_nc_def_dim(self._h, name.encode(), int(length or 0)))

    def def_var(self, name, nctype, dimids):
        d = (C.c_int * max(len(dimids), 1))(*dimids)
        return _nc_check(lib().fg_nc_def_var(self._h, name.encode(), int(nctype), len(dimids), d))

    def put_att(self, varid, name, value, nctype=NC_DOUBLE):
        if isinstance(value, str):
            _nc_check(lib().fg_nc_put_att_text(self._h, int(varid), name.encode(), value.encode()))
        else:
            v = np.atleast_1d(np.asarray(value, dtype=np.float64))
            _nc_check(lib().fg_nc_put_att_double(self._h, int(varid), name.encode(), int(nctype), len(v),
                                                 v.ctypes.data_as(C.POINTER(C.c_double))))

    def enddef(self):
        _nc_check(lib().fg_nc_enddef(self._h))

    # ---- inquiry (mpp_get_varid, mpp_get_var_ndim, mpp_get_var_att ...)
    @property
    def numrecs(self):
        return int(lib().fg_nc_inq_numrecs(self._h))

    def dims(self):
        out = {}
        for k in range(lib().fg_nc_inq_ndims(self._h)):
            nm = C.create_string_buffer(256); ln = C.c_long()
            _nc_check(lib().fg_nc_inq_dim(self._h, k, nm, 256, C.byref(ln)))
            out[nm.value.decode()] = int(ln.value)
        return out

    def varid(self, name):
        v = lib().fg_nc_inq_varid(self._h, name.encode())
        if v < 0:
            raise KeyError(name)
        return v

    def variables(self):
        return [self.inq_var(k)["name"] for k in range(lib().fg_nc_inq_nvars(self._h))]

    def inq_var(self, var):
        vid = self.varid(var) if isinstance(var, str) else int(var)
        nm = C.create_string_buffer(256); ty = C.c_int(); nd = C.c_int()
        dimids = (C.c_int * 32)(); shape = (C.c_long * 32)()
        _nc_check(lib().fg_nc_inq_var(self._h, vid, nm, 256, C.byref(ty), C.byref(nd), dimids, shape))
        return {"id": vid, "name": nm.value.decode(), "type": ty.value, "shape": tuple(shape[:nd.value]), "dimids": tuple(dimids[:nd.value])}

    def get_att(self, var, name, default=None):
        vid = -1 if var is None else (self.varid(var) if isinstance(var, str) else int(var))
        buf = (C.c_double * 64)()
        n = lib().fg_nc_get_att_double(self._h, vid, name.encode(), buf, 64)
        if n == FG_ERR_NOTFOUND:
            return default
        _nc_check(n)
        txt = C.create_string_buffer(4096)
        m = lib().fg_nc_get_att_text(self._h, vid, name.encode(), txt, 4096)
        if m >= 0:
            return txt.value.decode()
        return float(buf[0]) if n == 1 else np.array(buf[:n])

    # ---- data (mpp_get_var_value_block / mpp_put_var_value_block)
    def get_vara(self, var, start=None, count=None, as_double=False, out=None):
        info = self.inq_var(var)
        nd = len(info["shape"])
        start = [0] * nd if start is None else list(start)
        count = [info["shape"][d] - start[d] for d in range(nd)] if count is None else list(count)
        dt = np.float64 if as_double else _NP[info["type"]]
        n = int(np.prod(count)) if nd else 1
        if out is None:
            out = np.empty(n, dtype=dt)
        assert out.dtype == dt and out.size >= n and out.flags.c_contiguous
        s = (C.c_long * max(nd, 1))(*start); c = (C.c_long * max(nd, 1))(*count)
        fn = lib().fg_nc_get_vara_double if as_double else lib().fg_nc_get_vara
        _nc_check(fn(self._h, info["id"], s, c, out.ctypes.data_as(C.c_void_p)))
        return out[:n].reshape(count) if nd else out[:1]

    def put_vara(self, var, data, start=None, count=None):
        info = self.inq_var(var)
        nd = len(info["shape"])
        data = np.ascontiguousarray(data)
        start = [0] * nd if start is None else list(start)
        count = list(data.shape) if count is None else list(count)
        s = (C.c_long * max(nd, 1))(*start); c = (C.c_long * max(nd, 1))(*count)
        if data.dtype == _NP[info["type"]]:
            _nc_check(lib().fg_nc_put_vara(self._h, info["id"], s, c, data.ctypes.data_as(C.c_void_p)))
        else:
            d = np.ascontiguousarray(data, dtype=np.float64)
            _nc_check(lib().fg_nc_put_vara_double(self._h, info["id"], s, c, d.ctypes.data_as(C.c_void_p)))

    def close(self):
        if self._h:
            h, self._h = self._h, None
            _nc_check(lib().fg_nc_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_field_levels(files, name, level_t=None, level_n=None, kstart=0, nz=None, out=None):
    """get_input_data up to the host/device boundary (fregrid_util.c:2036-2123): the (t, [n], z-range, y, x) hyperslab of
    variable ``name`` of every tile file, in the FILE type, as one array [nz][sum of ny*nx] (tiles back to back, the layout
    the sweep takes), plus the variable's scale, offset and missing value as get_field_attribute reads them."""
    parts = []
    meta = None
    for f in files:
        info = f.inq_var(name)
        nd = len(info["shape"])
        start, count = [0] * nd, [1] * nd
        pos = 0
        if level_t is not None:
            start[pos] = level_t; pos += 1
        if level_n is not None:
            start[pos] = level_n; pos += 1
        if nd - pos == 3:                                 # has a z axis
            z = info["shape"][pos] - kstart if nz is None else nz
            start[pos] = kstart; count[pos] = z; pos += 1
        else:
            z = 1
        count[pos], count[pos + 1] = info["shape"][pos], info["shape"][pos + 1]
        a = f.get_vara(info["id"], start, count).reshape(z, -1)
        parts.append(a)
        if meta is None:
            meta = {"type": info["type"], "scale": f.get_att(name, "scale_factor", 0.0) or 0.0,
                    "offset": f.get_att(name, "add_offset", 0.0) or 0.0,
                    "missing": f.get_att(name, "missing_value", f.get_att(name, "_FillValue", None))}
    data = np.concatenate(parts, axis=1) if len(parts) > 1 else parts[0]
    if out is not None:
        out[:data.shape[0], :data.shape[1]] = data
        data = out[:data.shape[0]]
    return data, meta


class HostBuffer:
    """Page-locked host memory (fg_host_alloc) as a numpy array: the streamed sweep copies straight from / to it."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        lib().fg_host_alloc.restype = C.c_void_p
        self._p = lib().fg_host_alloc(C.c_size_t(max(nbytes, 1)))
        if not self._p:
            raise MemoryError("fg_host_alloc failed")
        buf = (C.c_char * max(nbytes, 1)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def free(self):
        if self._p:
            self.array = None
            lib().fg_host_free(C.c_void_p(self._p)); self._p = None


class Sweep:
    """fg_sweep: levels streamed from host memory through finalized plans (one per output tile) and back."""

    def __init__(self, plans, c2l=None, in_dtype=np.float64, out_dtype=np.float64):
        self.plans = list(plans)
        self.in_dtype, self.out_dtype = np.dtype(in_dtype), np.dtype(out_dtype)
        hs = (C.c_void_p * len(self.plans))(*[p._h for p in self.plans])
        self._h = C.c_void_p()
        check(lib().fg_sweep_create(len(self.plans), hs, c2l._h if c2l is not None else None, nc_type_of(self.in_dtype),
                                    nc_type_of(self.out_dtype), C.byref(self._h)))
        self.ncells_in = int(lib().fg_plan_ncells_in(self.plans[0]._h))
        self.ndst = [int(lib().fg_plan_ncells_out(p._h)) for p in self.plans]

    def run(self, host_in, outs, scale=0.0, offset=0.0, missing=-1.0e20, has_missing=False):
        """host_in [nlev][ncells_in] of in_dtype; outs[p] [nlev][ndst_p] of out_dtype (numpy arrays, ideally HostBuffer.array).
        `missing` only steers the scale / offset conversion (values equal to it are left alone, fregrid_util.c:2114-2123): the
        streamed remap itself treats every value as data.  A variable that HAS missing values (has_missing: its file carries
        missing_value / _FillValue, read_field_levels' meta["missing"] is not None) must go level by level through
        XgridPlan.apply(has_missing=True) -- conserve_interp.c:544 forbids nz > 1 for it -- so that case is refused here."""
        if has_missing and host_in.shape[0] > 1:
            raise ValueError("conserve_interp: has_missing should be false when nz > 1 (a variable with missing values cannot take the streamed sweep)")
        a = host_in
        assert a.dtype == self.in_dtype and a.flags.c_contiguous and a.shape[1] == self.ncells_in
        nlev = a.shape[0]
        for o, n in zip(outs, self.ndst):
            assert o.dtype == self.out_dtype and o.flags.c_contiguous and o.shape == (nlev, n)
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        check(lib().fg_sweep_run(self._h, a.ctypes.data_as(C.c_void_p), nlev, float(scale), float(offset), float(missing), ptrs))

    def destroy(self):
        if self._h:
            lib().fg_sweep_destroy(self._h); self._h = None
