"""Randomised A/B of the great-circle clip: three passes (k_gc_screen / k_gc_solve / k_gc_walk + list) against the one-kernel clip on
pairs of randomly rotated, randomly sized grids (lat-lon windows, cubed-sphere faces, tripolar) -- exchange cells and areas must be
bit-identical, or both must stop with the same reference error; small cases also against the brute-force CPU oracle.   usage: python scripts/gc_fuzz.py [cases] [seed]"""
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


def random_grid():
    kind = rng.integers(0, 4)
    if kind == 0:                                            # lat-lon window (or the whole sphere)
        nx, ny = int(rng.integers(8, 200)), int(rng.integers(6, 120))
        if rng.random() < 0.4:
            lo, la = fg.latlon_corners(nx, ny)
        else:
            l0, w = rng.uniform(0, 300), rng.uniform(20, 150); b0, h = rng.uniform(-80, 20), rng.uniform(15, 60)
            lo, la = fg.latlon_corners(nx, ny, l0, min(l0 + w, 359.0), b0, min(b0 + h, 89.0))
        g = (nx, ny, lo, la)
    elif kind == 1:                                          # one face of a cubed sphere
        n = int(rng.integers(6, 96)); c = fg.gnomonic_ed_corners(n); t = int(rng.integers(0, 6))
        g = (n, n, c[0][t], c[1][t])
    elif kind == 2:                                          # tripolar
        nx, ny = int(rng.integers(20, 160)), int(rng.integers(15, 100))
        g = (nx, ny) + tuple(fg.tripolar_corners(nx, ny))
    else:                                                    # fine lat-lon window
        nx, ny = int(rng.integers(100, 400)), int(rng.integers(60, 300))
        l0, b0 = rng.uniform(0, 340), rng.uniform(-70, 60)
        g = (nx, ny) + tuple(fg.latlon_corners(nx, ny, l0, l0 + rng.uniform(2, 15), b0, b0 + rng.uniform(2, 12)))
    if rng.random() < 0.6:
        g = g[:2] + rotated(g[2], g[3], rng.standard_normal(3), rng.uniform(0, np.pi))
    return g


def run(a, b, split):
    fg.lib().fg_set_gc_split(split)
    try:
        plan = fg.XgridPlan.create_great_circle([fg.GridConfig(*a)], fg.GridConfig(*b))
    except Exception as e:
        return None, str(e), {}
    x = plan.get_xgrid() if plan.nxgrid else {"area": np.zeros(0)}
    st = plan.stats()
    plan.destroy()
    return x, "", st


npairs = nx_tot = nerr = ndef = norc = nbit = ncmp = 0
for ci in range(ncase):
    a, b = random_grid(), random_grid()
    (x0, e0, s0), (x1, e1, s1) = run(a, b, 0), run(a, b, 1)
    assert e0 == e1, (ci, e0, e1)
    if x0 is None:
        nerr += 1
        continue
    assert len(x0["area"]) == len(x1["area"]), ci
    if len(x0["area"]):
        for k in ("i_in", "j_in", "i_out", "j_out"):
            assert np.array_equal(x0[k], x1[k]), (ci, k)
        assert np.array_equal(x0["area"].view(np.uint64), x1["area"].view(np.uint64)), ci
    npairs += s1["pairs"]; nx_tot += len(x1["area"]); ndef += s1["deferred"]
    if a[0] * a[1] * b[0] * b[1] <= 1_500_000:               # small enough for the brute-force CPU oracle (gc_oracle.c)
        o = orc.orc_create_xgrid_gc(a[0], a[1], b[0], b[1], a[2], a[3], b[2], b[3])
        assert o["n"] == len(x1["area"]), (ci, o["n"], len(x1["area"]))
        if o["n"]:
            for k in ("i_in", "j_in", "i_out", "j_out"):
                assert np.array_equal(x1[k], o[k]), (ci, k)
            rel = np.abs(x1["area"] - o["area"]) / o["area"]
            # The bar is 1e-10 relative -- or, for a small cell, the effect of ONE of its angles differing in the last place: the area
            # is (sum of angles - (n - 2) pi) R^2, so an ulp of an angle (2.2e-16 near pi/2) moves it by 0.009 m^2 whatever the cell's
            # size; the device's acosl equals glibc's x87 one except for 2.5e-4 of the arguments, by one ulp.  Eight such
            # ulps are allowed here (a polygon has up to 8 angles).
            tol = np.maximum(1e-10 * o["area"], 8 * 2.220446049250313e-16 * 6371000.0 ** 2)
            rel = np.where(np.abs(x1["area"] - o["area"]) <= tol, 0.0, rel)
            if not np.max(rel) < 1e-10:                       # say which cell, how thin, by how much -- before failing
                k = int(np.argmax(rel))
                big = float(np.max(o["area"]))
                print(f"case {ci}: area mismatch at exchange cell {k}: device {x1['area'][k]!r} oracle {o['area'][k]!r} rel {rel[k]:.3e}; "
                      f"cell / largest cell of the pair of grids {o['area'][k] / big:.3e}; {int(np.sum(rel >= 1e-10))} of {o['n']} cells beyond 1e-10", flush=True)
            assert np.max(rel) < 1e-10, ci
            nbit += int(np.sum(x1["area"].view(np.uint64) == o["area"].view(np.uint64))); ncmp += o["n"]
        norc += 1
    print(f"case {ci}: {a[0]}x{a[1]} vs {b[0]}x{b[1]}: pairs {s1['pairs']}, nxgrid {len(x1['area'])}, listed {s1['deferred']}", flush=True)
fg.lib().fg_set_gc_split(1)
print(f"gc_fuzz: {ncase} grid pairs, {ncase - nerr} clipped ({nerr} stopped by the same reference error in both), "
      f"{npairs} candidate pairs, {nx_tot} exchange cells, {ndef} pairs through the list: all bit-identical; {norc} pairs of grids also "
      f"against the CPU oracle: lists identical, areas within 1e-10, {nbit} of {ncmp} areas bit-identical")
