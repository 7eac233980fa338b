#!/usr/bin/env python3
"""Golden vectors of the great-circle path from the COMPILED REFERENCE (oracle/_ref): create_xgrid_great_circle
(tools/libfrencutils/create_xgrid.c:1366) for C24 gnomonic_ed tiles 1 and 3 (polar) against the 144x90 lat-lon
grid, and get_grid_great_circle_area of those tiles.  Run in the build container:  python tests/golden/make_golden_gc.py
Output: gc_c24_xgrid.npz (inputs = the corner arrays; outputs = the reference's index lists and areas)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gridutil  # noqa: E402
import orc  # noqa: E402


def main():
    assert orc.ref_available(), "build oracle/_ref first (make -C oracle)"
    lon, lat = gridutil.ref_gnomonic_corners(24)
    D2R = np.pi / 180
    nlon, nlat = 144, 90
    lo1 = np.linspace(0.0, 360.0, nlon + 1) * D2R
    la1 = np.linspace(-90.0, 90.0, nlat + 1) * D2R
    lo, la = np.meshgrid(lo1, la1)
    out = dict(lon_out=lo, lat_out=la)
    for t in (0, 2):
        r = orc.ref_create_xgrid_gc(24, 24, nlon, nlat, lon[t], lat[t], lo, la)
        out[f"lon_t{t}"], out[f"lat_t{t}"] = lon[t], lat[t]
        for k in ("i_in", "j_in", "i_out", "j_out", "area"):
            out[f"{k}_t{t}"] = r[k]
        out[f"cell_area_t{t}"] = orc.ref_get_grid_gc_area(24, 24, lon[t], lat[t])
        print("tile", t + 1, "nxgrid", r["n"])
    np.savez_compressed(os.path.join(HERE, "gc_c24_xgrid.npz"), **out)


if __name__ == "__main__":
    main()
