"""CPU tests for the order-2 input preparation: host geometry (calc_c2l_grid_info restatement, cell
centres, tile contacts, halo gather map) against the compiled reference where it can be compiled, and the
gradient oracle against the compiled reference."""
import ctypes as C

import numpy as np
import pytest

import gridutil
import orc

dp, ip = orc.dp, orc.ip
P = lambda a: a.ctypes.data_as(dp)
PI = lambda a: a.ctypes.data_as(ip)
needs_ref = pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def halo_centres(fg, ni, lont, latt, contacts):
    off, m = fg.halo_map([ni] * 6, [ni] * 6, contacts)
    F = int(off[-1])
    xt, yt = np.zeros(F), np.zeros(F)
    for t in range(6):
        v = xt[off[t]:off[t + 1]].reshape(ni + 2, ni + 2); v[1:-1, 1:-1] = lont[t]
        v = yt[off[t]:off[t + 1]].reshape(ni + 2, ni + 2); v[1:-1, 1:-1] = latt[t]
    g = m >= 0
    xt[g] = xt[m[g]]; yt[g] = yt[m[g]]
    return off, m, xt, yt


@needs_ref
def test_cell_centres_match_reference_generator(fg):
    for ni in (8, 48):
        lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
        rl, ra, rlt, rat = gridutil.ref_gnomonic_corners(ni, centres=True)
        assert np.array_equal(_bits(lon), _bits(rl)) and np.array_equal(_bits(lat), _bits(ra))
        assert np.array_equal(_bits(lont), _bits(rlt)) and np.array_equal(_bits(latt), _bits(rat))


@needs_ref
def test_contacts_match_reference_get_align_contact(fg):
    """make_solo_mosaic's get_align_contact on the supergrid (get_contact.c:44) + the supergrid -> model index rule of
    read_mosaic_contact (read_mosaic.c:655-681, restated below) must give the contacts our geometric search finds."""
    R = orc.ref()
    ni = 8
    x, y = gridutil.ref_gnomonic_corners(ni, supergrid=True)
    nxp = 2 * ni + 1
    f = R.get_align_contact
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 6 + [dp] * 4 + [C.c_double, C.c_double] + [ip] * 8

    def to_model(a, b, refine=2):
        if a == b:
            v = (a + 1) // refine - 1
            return v, v
        if b > a:
            s, e = a - 1, b - refine
        else:
            s, e = a - refine, b - 1
        assert s % refine == 0 and e % refine == 0
        return s // refine, e // refine

    ref = set()
    for n in range(6):
        for m in range(n, 6):
            arr = [np.zeros(8, dtype=np.int32) for _ in range(8)]
            xs = [np.ascontiguousarray(v) for v in (x[n], y[n], x[m], y[m])]
            cnt = f(n + 1, m + 1, nxp, nxp, nxp, nxp, P(xs[0]), P(xs[1]), P(xs[2]), P(xs[3]), 0.0, 0.0, *[PI(a) for a in arr])
            for k in range(cnt):
                i1 = to_model(int(arr[0][k]), int(arr[1][k])); j1 = to_model(int(arr[2][k]), int(arr[3][k]))
                i2 = to_model(int(arr[4][k]), int(arr[5][k])); j2 = to_model(int(arr[6][k]), int(arr[7][k]))
                ref.add((n + 1, i1, j1, m + 1, i2, j2))
    lon, lat = fg.gnomonic_ed_corners(ni)
    c = fg.find_contacts([ni] * 6, [ni] * 6, lon, lat)
    assert len(c["tile1"]) == len(ref) == 12

    def norm(t1, i1, j1, t2, i2, j2):
        # orientation may be carried by either side; normalise: side 1 ascending
        if i1[0] > i1[1] or j1[0] > j1[1]:
            i1, j1 = tuple(sorted(i1)), tuple(sorted(j1))
            i2, j2 = i2[::-1] if i2[0] != i2[1] else i2, j2[::-1] if j2[0] != j2[1] else j2
        return (t1, tuple(i1), tuple(j1), t2, tuple(i2), tuple(j2))

    ours = {norm(int(c["tile1"][k]), (int(c["istart1"][k]), int(c["iend1"][k])), (int(c["jstart1"][k]), int(c["jend1"][k])),
                 int(c["tile2"][k]), (int(c["istart2"][k]), int(c["iend2"][k])), (int(c["jstart2"][k]), int(c["jend2"][k])))
            for k in range(12)}
    assert ours == {norm(*r) for r in ref}


def test_halo_map_equals_reference_form_and_is_geometrically_continuous(fg):
    """The folded gather map (product) against the two-step Bound-table restatement (oracle), and a geometric
    property: a halo cell centre must be the neighbour tile's cell centre adjacent across the edge, i.e. about one
    grid spacing beyond the edge cell and NOT inside this tile."""
    L = orc.oracle()
    ni = 12
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    c = fg.find_contacts([ni] * 6, [ni] * 6, lon, lat)
    off, m, xt, yt = halo_centres(fg, ni, lont, latt, c)
    rng = np.random.default_rng(5)
    nz = 2
    tiles = [np.zeros((nz, ni + 2, ni + 2)) for _ in range(6)]
    for t in range(6):
        tiles[t][:, 1:-1, 1:-1] = rng.standard_normal((nz, ni, ni))
    mine = [a.copy() for a in tiles]
    flat = np.concatenate([a.reshape(nz, -1) for a in mine], axis=1)
    g = m >= 0
    flat[:, g] = flat[:, m[g]]
    f = L.orc_update_halo
    f.restype = C.c_int
    f.argtypes = [C.c_int, ip, ip, C.c_int] + [ip] * 10 + [C.c_int, C.POINTER(dp)]
    nxa = np.full(6, ni, dtype=np.int32)
    keys = ["tile1", "tile2", "istart1", "iend1", "jstart1", "jend1", "istart2", "iend2", "jstart2", "jend2"]
    ptrs = (dp * 6)(*[P(a) for a in tiles])
    assert f(6, PI(nxa), PI(nxa), 12, *[PI(np.ascontiguousarray(c[k])) for k in keys], nz, ptrs) == 0
    ref_flat = np.concatenate([a.reshape(nz, -1) for a in tiles], axis=1)
    assert np.array_equal(_bits(flat), _bits(ref_flat))
    # every non-corner halo cell is filled
    for t in range(6):
        mm = m[off[t]:off[t + 1]].reshape(ni + 2, ni + 2)
        assert np.all(mm[0, 1:-1] >= 0) and np.all(mm[-1, 1:-1] >= 0) and np.all(mm[1:-1, 0] >= 0) and np.all(mm[1:-1, -1] >= 0)
        assert np.all(mm[1:-1, 1:-1] < 0) and mm[0, 0] < 0 and mm[-1, -1] < 0
    # geometric continuity of the halo'd centres
    def xyz(lo, la):
        return np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)], axis=-1)
    for t in range(6):
        X = xyz(xt[off[t]:off[t + 1]].reshape(ni + 2, ni + 2), yt[off[t]:off[t + 1]].reshape(ni + 2, ni + 2))
        for (h, e, e2) in ((X[0, 1:-1], X[1, 1:-1], X[2, 1:-1]), (X[-1, 1:-1], X[-2, 1:-1], X[-3, 1:-1]),
                           (X[1:-1, 0], X[1:-1, 1], X[1:-1, 2]), (X[1:-1, -1], X[1:-1, -2], X[1:-1, -3])):
            d_he = np.linalg.norm(h - e, axis=-1)          # halo -> edge cell
            d_ee = np.linalg.norm(e - e2, axis=-1)         # edge cell -> next interior cell
            d_h2 = np.linalg.norm(h - e2, axis=-1)
            assert np.all(d_he > 0.5 * d_ee) and np.all(d_he < 1.6 * d_ee)
            assert np.all(d_h2 > 1.4 * d_ee)               # the halo point lies beyond the edge, not back inside


@needs_ref
def test_c2l_grid_info_bitwise_vs_reference(fg):
    R = orc.ref()
    ni = 24
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    c = fg.find_contacts([ni] * 6, [ni] * 6, lon, lat)
    off, m, xt, yt = halo_centres(fg, ni, lont, latt, c)
    f = R.calc_c2l_grid_info
    f.restype = None
    f.argtypes = [ip, ip] + [dp] * 15 + [ip] * 4
    for t in (0, 2, 5):
        mine = fg.c2l_grid_info(ni, ni, xt[off[t]:off[t + 1]], yt[off[t]:off[t + 1]], lon[t], lat[t])
        ref = {k: np.empty_like(v) for k, v in mine.items()}
        one = C.c_int(1)
        n = C.c_int(ni)
        xs = [np.ascontiguousarray(v) for v in (xt[off[t]:off[t + 1]], yt[off[t]:off[t + 1]], lon[t].ravel(), lat[t].ravel())]
        f(C.byref(n), C.byref(n), *[P(v) for v in xs],
          *[P(ref[k]) for k in ("dx", "dy", "area", "edge_w", "edge_e", "edge_s", "edge_n", "en_n", "en_e", "vlon", "vlat")],
          C.byref(one), C.byref(one), C.byref(one), C.byref(one))
        for k in mine:
            assert np.array_equal(_bits(mine[k]), _bits(ref[k])), (t, k)


@needs_ref
def test_grad_oracle_bitwise_vs_reference(fg):
    L, R = orc.oracle(), orc.ref()
    ni = 16
    lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
    c = fg.find_contacts([ni] * 6, [ni] * 6, lon, lat)
    off, m, xt, yt = halo_centres(fg, ni, lont, latt, c)
    g = L.orc_grad_c2l
    g.restype = None
    g.argtypes = [C.c_int, C.c_int] + [dp] * 14
    r = R.grad_c2l
    r.restype = None
    r.argtypes = [ip, ip] + [dp] * 14 + [ip] * 4
    rng = np.random.default_rng(11)
    for t in (0, 2):
        info = fg.c2l_grid_info(ni, ni, xt[off[t]:off[t + 1]], yt[off[t]:off[t + 1]], lon[t], lat[t])
        pin = rng.standard_normal((ni + 2) * (ni + 2)) + 3.0
        keys = ("dx", "dy", "area", "edge_w", "edge_e", "edge_s", "edge_n", "en_n", "en_e", "vlon", "vlat")
        gx1, gy1, gx2, gy2 = (np.empty(ni * ni) for _ in range(4))
        g(ni, ni, P(pin), *[P(info[k]) for k in keys], P(gx1), P(gy1))
        n, one = C.c_int(ni), C.c_int(1)
        r(C.byref(n), C.byref(n), P(pin), *[P(info[k]) for k in keys], P(gx2), P(gy2),
          C.byref(one), C.byref(one), C.byref(one), C.byref(one))
        assert np.array_equal(_bits(gx1), _bits(gx2)) and np.array_equal(_bits(gy1), _bits(gy2))
