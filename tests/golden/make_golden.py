#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE (oracle/_ref).

Run in the build container (needs /root/reference via oracle/Makefile):  python tests/golden/make_golden.py
Outputs (data only -- inputs and the reference's outputs):
  clip_cases.json      the legacy-clip known-answer cases 15..26 embedded in the reference's test main
                       (tools/libfrencutils/create_xgrid.c:2825-3015; expectations :3125-3130), inputs in
                       degrees as written there, outputs of the reference's clip_2dx2d/fix_lon/poly_area
  c48_xgrid.npz        C48 gnomonic_ed tiles 1 and 3 (polar): create_xgrid_2dx2d_order1 vs 180x90 and
                       create_xgrid_2dx2d_order2 vs 144x90 (the tests/fregrid/cubedsphere target), full lists
  c48_grid.npz         the C48 corner arrays those runs used (reference generator, create_gnomonic_cubic_grid.c:101)
  counts.json          nxgrid counts measured with the reference for larger cases
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gridutil  # noqa: E402
import orc  # noqa: E402

D2R = np.pi / 180

# (case, lon1, lat1, lon2, lat2, expectation text at create_xgrid.c:3125-3130)
CASES = [
    (15, [145.159, 198.302, 262.400, 262.400, 82.400, 82.400], [89.642, 89.648, 89.847, 90.000, 90.000, 89.835],
         [150.000, 177.824, 240.000, 240.000, 150.000], [89.789, 89.761, 89.889, 90.000, 90.000], "second box"),
    (16, [82.400, 82.400, 262.400, 262.400, 326.498, 379.641], [89.835, 90.000, 90.000, 89.847, 89.648, 89.642],
         [302.252, 330.000, 330.000, 240.000, 240.000], [89.876, 89.891, 90.000, 90.000, 89.942], "n_out=5"),
    (17, [82.400, 82.400, 262.400, 262.400, 326.498, 379.641], [89.835, 90.000, 90.000, 89.847, 89.648, 89.642],
         [-30.000, -2.252, 60.000, 60.000, -30.000], [89.891, 89.876, 89.942, 90.000, 90.000], "second box"),
    (18, [82.400, 82.400, 262.400, 262.400, 326.498, 379.641], [89.835, 90.000, 90.000, 89.847, 89.648, 89.642],
         [150.000, 177.824, 240.000, 240.000, 150.000], [89.789, 89.761, 89.889, 90.000, 90.000], "n_out=0"),
    (19, [145.159, 198.302, 262.400, 262.400, 82.400, 82.400], [89.642, 89.648, 89.847, 90.000, 90.000, 89.835],
         [-30.000, -2.176, 60.000, 60.000, -30.000], [89.789, 89.761, 89.889, 90.000, 90.000], "n_out=0"),
    (20, [145.159, 198.302, 262.400, 262.400, 82.400, 82.400], [89.642, 89.648, 89.847, 90.000, 90.000, 89.835],
         [122.176, 150.000, 150.000, 60.000, 60.000], [89.761, 89.789, 90.000, 90.000, 89.889], "n_out=5"),
    (21, [82.400, 82.400, 262.400, 262.400, 326.498, 379.641], [89.835, 90.000, 90.000, 89.847, 89.648, 89.642],
         [122.176, 150.000, 150.000, 60.000, 60.000], [89.761, 89.789, 90.000, 90.000, 89.889], "n_out=4"),
    (26, [209.68793552504, 158.60256162113, 82.40000000000, 262.40000000000],
         [-89.11514201451, -89.26896927380, -89.82370183256, -89.46584623220], None, None, "same box; area24=25=26"),
    (23, [158.60256162113, 121.19651597620, 82.40000000000, 82.40000000000],
         [-89.26896927380, -88.85737639760, -89.10746816044, -89.82370183256], None, None, "same box; area22=23"),
    (24, [262.40000000000, 262.40000000000, 82.4, 82.4, 6.19743837887, -44.88793552504],
         [-89.46584623220, -90.0, -90.0, -89.82370183256, -89.26896927380, -89.11514201451], None, None, "same box; area24=25=26"),
    (25, [262.40000000000, 82.4, 6.19743837887, -44.88793552504],
         [-89.46584623220, -89.82370183256, -89.26896927380, -89.11514201451], None, None, "same box; area24=25=26"),
    (22, [82.4, 82.4, 43.60348402380, 6.19743837887],
         [-89.82370183256, -89.10746816044, -88.85737639760, -89.26896927380], None, None, "same box; area22=23"),
]


def run_case(R, lon1, lat1, lon2, lat2):
    """What the reference's test main does for n > 14 (create_xgrid.c:3092-3104)."""
    dp = orc.dp
    a = lambda v: np.array(list(v) + [0.0] * (60 - len(v)), dtype=np.float64)
    x1, y1, x2, y2 = a(np.array(lon1) * D2R), a(np.array(lat1) * D2R), a(np.array(lon2) * D2R), a(np.array(lat2) * D2R)
    xo, yo = np.zeros(60), np.zeros(60)
    P = lambda v: v.ctypes.data_as(dp)
    n_clip = R.clip_2dx2d(P(x1), P(y1), len(lon1), P(x2), P(y2), len(lon2), P(xo), P(yo))
    clip_x, clip_y = xo[:n_clip].copy(), yo[:n_clip].copy()
    n1 = R.fix_lon(P(x1), P(y1), len(lon1), np.pi)
    n2 = R.fix_lon(P(x2), P(y2), len(lon2), np.pi)
    n_out = R.fix_lon(P(xo), P(yo), n_clip, np.pi)
    return dict(n_clip=int(n_clip), clip_lon=clip_x.tolist(), clip_lat=clip_y.tolist(),
                n1_fixed=int(n1), n2_fixed=int(n2), n_out_fixed=int(n_out),
                out_lon=xo[:n_out].tolist(), out_lat=yo[:n_out].tolist(),
                area1=R.poly_area(P(x1), P(y1), n1), area2=R.poly_area(P(x2), P(y2), n2),
                area_out=R.poly_area(P(xo), P(yo), n_out))


def main():
    R = orc.ref()
    if R is None:
        raise SystemExit("oracle/_ref/libfrenc_ref.so missing: run `make -C oracle` where /root/reference exists")
    cases = []
    for (n, lon1, lat1, lon2, lat2, expect) in CASES:
        if lon2 is None:
            lon2, lat2 = lon1, lat1
        r = run_case(R, lon1, lat1, lon2, lat2)
        cases.append(dict(case=n, lon1_deg=lon1, lat1_deg=lat1, lon2_deg=lon2, lat2_deg=lat2, expect=expect, ref=r))
    json.dump(dict(source="tools/libfrencutils/create_xgrid.c:2825-3015,3092-3130 run through oracle/_ref", cases=cases),
              open(os.path.join(HERE, "clip_cases.json"), "w"), indent=1)

    lon, lat = gridutil.ref_gnomonic_corners(48)
    np.savez_compressed(os.path.join(HERE, "c48_grid.npz"), lon=lon, lat=lat)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from conftest import load_package
    fg = load_package()
    out = {}
    for order, nlon, nlat in ((1, 180, 90), (2, 144, 90)):
        lo, la = fg.latlon_corners(nlon, nlat)
        for t in (0, 2):
            r = orc.ref_create_xgrid(order, 48, 48, nlon, nlat, lon[t], lat[t], lo, la)
            key = f"o{order}_{nlon}x{nlat}_t{t + 1}"
            for k, v in r.items():
                if k != "n":
                    out[f"{key}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "c48_xgrid.npz"), **out)
    counts = {
        "source": "BASELINE.md §2: unmodified reference, gnomonic_ed (shift_fac=18) -> global regular lat-lon, mask=1",
        "C48->180x90 o1": {"total": 63752, "tiles_1245": 8460, "tiles_36": 14956},
        "C48->144x90 o2": {"total": 55904, "tiles_1245": 7584, "tiles_36": 12784},
        "C96->360x180 o2": {"total": 256864, "tiles_1245": 33972, "tiles_36": 60488},
        "C192->720x360 o2": {"total": 1035968},
        "C384->1440x720 o2": {"total": 4160000, "tiles_1245": 550036, "tiles_36": 979928,
                              "first_xcell_tile1": [0, 0, 1220, 218], "first_area_tile1": 118553835.02491695},
    }
    json.dump(counts, open(os.path.join(HERE, "counts.json"), "w"), indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
