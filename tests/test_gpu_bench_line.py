"""bench.py prints ONE JSON line with the keys the driver and the judge read (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, plus roofline and cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema(gpu_ok):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "2", "--apply-steps", "5",
                        "--legs", "cpu", "--cpu-rows", "1", "--gc-steps", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"] == d["unit"] == "exchange-cells/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None
    assert "C384" in d["config"]["workload"] and d["config"]["nxgrid"] == 4160000      # the reference's own count
    assert abs(d["value"] - 3 * 4160000 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert 0.5 < d["ms_per_step"] < 5.0
    rf = d["roofline"]
    assert rf["bound"] == "valu" and rf["unit"] and 0 < rf["frac"] <= 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None or rf["traffic"] > 0
    ra = d["roofline_apply"]
    assert ra["bound"] == "hbm" and ra["unit"] == "GB/s" and ra["peak"] == 8000.0 and 0 < ra["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "exchange-cells/s" and cb["sample"]
    assert abs(d["mass_rel_err"]) < 2e-9 and abs(d["mass_rel_err_xgrid"]) < 1e-13


def test_bench_two_ranks_rehearsal(gpu_ok):
    """`bench.py --gpus 2` the way the driver launches it (torch.distributed.run, one process per rank), but with both ranks on
    this one GPU and gloo for the collectives (FG_BENCH_BACKEND=gloo): times mean nothing, the N > 1 code path does -- the
    banded, culled search, the rank-to-rank hand-over of the shared cells' sums, the sweep of a band, the C768 jobs (legacy and great circle).  (This rehearsal found a fault and a
    wrong flux sum that no single-process test could see.)"""
    env = dict(os.environ, FG_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "1",
                        "--apply-steps", "5"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["scaling"] == "strong"
    assert d["config"]["nxgrid"] == 4160000                           # the two bands together: the single-rank count
    assert abs(d["mass_rel_err"]) < 2e-9 and abs(d["mass_rel_err_xgrid"]) < 1e-13
    assert d["c768_order2"]["nxgrid"] == 16673872 and d["c768_order2"]["n_gpus"] == 2
    assert 0 < d["roofline"]["frac"] <= 1 and d["value"] > 0
    ck = d["exchange_check"]                                         # the timed steps' hand-over carries the single-rank sums' bits
    assert ck["bit_identical_to_single_rank_sums"] is True and ck["cells_on_this_rank"] > 400000
    # (two bands meet on the equator, a grid line of the cubed sphere: no source cell is shared; tests/test_gpu_multirank.py hands cells over on 3 ranks)
    assert d["ms_per_step_allreduce_exchange"] > 0
    g = d["c768_great_circle"]                                       # BASELINE config 4: great circle under ranks + gathered remap file
    assert g["nxgrid"] == 16674181 and g["n_gpus"] == 2 and g["remap_write_s"] > 0 and g["remap_file_bytes"] > 16674181 * 28
