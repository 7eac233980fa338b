"""Time of the per-level gradient preparation (get_input_data's device part: halo update + grad_c2l) for nz = 1 and 8, C384."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
fg = ge.load_package()
ni = 384
lon, lat, lont, latt = fg.gnomonic_ed_grid(ni)
prep = fg.C2lPrep([ni] * 6, [ni] * 6, lon, lat, lont, latt, fg.find_contacts([ni] * 6, [ni] * 6, lon, lat), device=0)
dev = "cuda:0"
nc = 6 * ni * ni
rng = np.random.default_rng(0)
for nz in (1, 8):
    src = torch.from_numpy(rng.standard_normal((nz, nc))).to(dev)
    srcm = src.clone(); srcm[torch.rand_like(srcm) < 0.2] = -1.0e10
    halo = torch.empty(nz, prep.F, dtype=torch.float64, device=dev)
    gx = torch.empty(nz, nc, dtype=torch.float64, device=dev); gy = torch.empty_like(gx)
    gm = torch.empty(nc, dtype=torch.int32, device=dev)
    def run(f, label):
        for _ in range(5): f()
        prep.sync(); t0 = time.perf_counter()
        for _ in range(100): f()
        prep.sync(); dt = (time.perf_counter() - t0) / 100
        print(f"nz {nz} {label}: {dt * 1e3:.4f} ms per call", flush=True)
    run(lambda: prep.fill_halo(src, halo, nz), "fill_halo")
    run(lambda: prep.gradient(halo, nz, gx, gy), "gradient")
    if nz == 1:
        prep.fill_halo(srcm, halo, 1)
        run(lambda: prep.gradient(halo, 1, gx, gy, gm, True, -1.0e10), "gradient with missing values + mask")
