"""Randomised check of the legacy (lon-lat plane) search, order 2: the single-sync search (default: the rectilinear-target path when
the target is a lat-lon grid, global / regional / stretched / on -180..180) against the exactly sized one (fg_set_search_mode(1)),
against the GENERIC bins path (fg_set_search_rect(0)), against the generic path in 3 chunks of source cells and against a culling
search, on pairs of random grids (lat-lon windows and global grids, cubed-sphere
faces, tripolar, rotated versions of them); small cases also against the CPU oracle (lists identical, areas / centroid integrals bit
for bit where the host libm matches).  Both sides stopping with the same reference error counts as agreement.
usage: python scripts/legacy_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_package
import orc
fg = load_package()
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rotated(lon, lat, ax, ang):
    x, y, z = np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)
    v = np.stack([x, y, z], -1)
    ax = np.asarray(ax, float) / np.linalg.norm(ax)
    c, s = np.cos(ang), np.sin(ang)
    r = v * c + np.cross(ax, v) * s + ax * (v @ ax)[..., None] * (1 - c)
    lo = np.arctan2(r[..., 1], r[..., 0]); lo[lo < 0] += 2 * np.pi
    return np.ascontiguousarray(lo), np.ascontiguousarray(np.arcsin(np.clip(r[..., 2], -1, 1)))


def random_grid(small):
    kind = rng.integers(0, 3)
    top = 40 if small else 160
    if kind == 0:
        nx, ny = int(rng.integers(6, top)), int(rng.integers(4, max(6, top // 2)))
        u = rng.random()
        if u < 0.35:
            lo, la = fg.latlon_corners(nx, ny)
        elif u < 0.5:
            lo, la = fg.latlon_corners(nx, ny, -180.0, 180.0, -90.0, 90.0)
        else:
            l0, w = rng.uniform(-170, 300), rng.uniform(20, 150); b0, h = rng.uniform(-80, 20), rng.uniform(15, 60)
            lo, la = fg.latlon_corners(nx, ny, l0, min(l0 + w, 359.0), b0, min(b0 + h, 89.0))
        if rng.random() < 0.25:                              # stretched axes: rectilinear, not uniform
            i, j = np.arange(nx + 1) / nx, np.arange(ny + 1) / ny
            ax = lo[0, 0] + (lo[0, -1] - lo[0, 0]) * i ** rng.uniform(0.6, 1.8)
            ay = la[0, 0] + (la[-1, 0] - la[0, 0]) * (0.5 - 0.5 * np.cos(np.pi * j))
            lo = np.ascontiguousarray(np.broadcast_to(ax[None, :], lo.shape)); la = np.ascontiguousarray(np.broadcast_to(ay[:, None], la.shape))
        g = (nx, ny, lo, la)
    elif kind == 1:
        n = int(rng.integers(4, max(6, top // 2))); c = fg.gnomonic_ed_corners(n); t = int(rng.integers(0, 6))
        g = (n, n, c[0][t], c[1][t])
    else:
        nx, ny = int(rng.integers(12, top)), int(rng.integers(10, max(12, top // 2)))
        g = (nx, ny) + tuple(fg.tripolar_corners(nx, ny))
    if rng.random() < 0.3:                                   # a rotated grid away from the poles of the rotation: still quads in the lon-lat plane
        lo, la = rotated(g[2], g[3], rng.standard_normal(3), rng.uniform(0, 0.4))
        if np.abs(la).max() < 1.45 and lo.min() > 0.3 and lo.max() < 5.9:
            g = g[:2] + (lo, la)
    return g


def run(a, b, mode, chunks, rect=1, cull=0):
    fg.lib().fg_set_search_mode(mode); fg.lib().fg_set_search_chunks(chunks); fg.lib().fg_set_search_rect(rect); fg.lib().fg_set_search_cull(cull)
    try:
        plan = fg.XgridPlan.create(2, [fg.GridConfig(*a)], fg.GridConfig(*b))
        plan.finalize()
    except Exception as e:
        return None, str(e), {}
    x = plan.get_xgrid() if plan.nxgrid else {"area": np.zeros(0)}
    st = plan.stats()
    plan.destroy()
    return x, "", st


bits = lambda v: np.ascontiguousarray(v).view(np.uint64)
nx_tot = nerr = norc = nrect = 0
try:
    for ci in range(ncase):
        small = rng.random() < 0.5
        a, b = random_grid(small), random_grid(small)
        res = [run(a, b, 0, 0), run(a, b, 1, 0), run(a, b, 0, 0, rect=0), run(a, b, 0, 3, rect=0), run(a, b, 0, 0, cull=1)]
        x0, e0, s0 = res[0]
        for x, e, s in res[1:]:
            assert e == e0, (ci, e0, e)
            if x0 is None:
                continue
            assert len(x["area"]) == len(x0["area"]), ci
            if len(x0["area"]):
                for k in ("i_in", "j_in", "i_out", "j_out"):
                    assert np.array_equal(x0[k], x[k]), (ci, k)
                for k in ("area", "c1", "c2"):
                    assert np.array_equal(bits(x0[k]), bits(x[k])), (ci, k)
        if x0 is None:
            nerr += 1
            print(f"case {ci}: stopped: {e0[:80]}", flush=True)
            continue
        nx_tot += len(x0["area"])
        nrect += 1 if s0["bins"] == 0 else 0
        tag = " (rectilinear path)" if s0["bins"] == 0 else ""
        if small and a[0] * a[1] * b[0] * b[1] <= 3_000_000:
            try:
                o = orc.orc_setup(2, [a], [b])
            except Exception as e:
                o = None
            if o is not None:
                assert o["n"] == len(x0["area"]), (ci, o["n"], len(x0["area"]))
                if o["n"]:
                    for k in ("i_in", "j_in", "i_out", "j_out"):
                        assert np.array_equal(x0[k], o[k]), (ci, k)
                    assert np.max(np.abs(x0["area"] - o["area"]) / o["area"]) < 1e-10, ci
                    if orc.host_has_fma():
                        assert np.array_equal(bits(x0["area"]), bits(o["area"])), ci
                        assert np.array_equal(bits(x0["c1"]), bits(o["di"])) and np.array_equal(bits(x0["c2"]), bits(o["dj"])), ci
                norc += 1
                tag += " = oracle"
        print(f"case {ci}: {a[0]}x{a[1]} vs {b[0]}x{b[1]}: nxgrid {len(x0['area'])}, pairs {s0['pairs']}, exact_mode {s0['exact_mode']}{tag}", flush=True)
finally:
    fg.lib().fg_set_search_mode(0); fg.lib().fg_set_search_chunks(0); fg.lib().fg_set_search_rect(1); fg.lib().fg_set_search_cull(0)
print(f"legacy_fuzz: {ncase} grid pairs, {ncase - nerr} searched ({nerr} stopped by the same reference error in all variants), {nx_tot} exchange cells, "
      f"default == exactly sized == generic path == generic in 3 chunks == culling search bit for bit ({nrect} targets took the rectilinear path); "
      f"{norc} of them also equal to the CPU oracle")
