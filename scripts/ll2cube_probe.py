"""Phase times of lat-lon -> cubed-sphere tile searches (generic path: curvilinear target).  usage: ll2cube_probe.py [nlon nlat ni]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
a = [int(v) for v in sys.argv[1:]]
nlon, nlat, ni = (a + [1440, 720, 384][len(a):])[:3]
lo, la = fg.latlon_corners(nlon, nlat); lon, lat = fg.gnomonic_ed_corners(ni)
dev = "cuda:0"
h2d = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
lo_t, la_t = [h2d(lo)], [h2d(la)]
fg.lib().fg_set_profiling(1)
for cull in (0, 1):
    fg.lib().fg_set_search_cull(cull)
    for t in (0, 2):
        dl, da = h2d(lon[t]), h2d(lat[t])
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            p = fg.XgridPlan.create_dev(1, [nlon], [nlat], lo_t, la_t, ni, ni, dl, da, 0.0, 0.0)
            p.finalize(); p.sync(); dt = time.perf_counter() - t0
            ph = p.phase_ms(); st = p.stats(); p.destroy()
        print(f"cull {cull} tile {t + 1}: {dt * 1e3:.3f} ms nxgrid {st['nxgrid']} pairs {st['pairs']} deferred {st['deferred']} heavy {st['heavy']} bins {st['bins']} entries {st['bin_entries']}",
              {k: round(v, 3) for k, v in ph.items() if v > 0}, flush=True)
fg.lib().fg_set_search_cull(0)
