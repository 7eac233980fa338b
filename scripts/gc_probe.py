"""Great-circle search + finalize a few times for rocprofv3 --stats.  usage: gc_probe.py ni nlon nlat"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
ni, nlon, nlat = (int(v) for v in sys.argv[1:4])
lon, lat = fg.gnomonic_ed_corners(ni); lo, la = fg.latlon_corners(nlon, nlat)
h2d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
xin = [tuple(h2d(a) for a in fg.latlon2xyz(lon[t], lat[t])) for t in range(6)]
xo = tuple(h2d(a) for a in fg.latlon2xyz(lo, la))
for _ in range(6):
    p = fg.XgridPlan.create_great_circle_dev([ni] * 6, [ni] * 6, xin, nlon, nlat, xo, np.pi / nlat, 2 * np.pi / nlon)
    p.finalize(); p.sync(); p.destroy()
