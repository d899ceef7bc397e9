"""CPU: the parts of bench.py that do not need a GPU -- which time slabs the parity leg checks,
when a committed counter profile may be quoted, and the P-process CPU baseline's plumbing."""

import importlib.util
import json
import os

import numpy as np

from momlevel_amd import synthetic
from oracle import momlevel_numpy as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_parity_slabs_reach_every_time_chunk():
    assert bench.parity_slabs(120) == [0, 32, 64, 96, 31, 63, 95, 119]
    assert bench.parity_slabs(1) == [0]
    assert bench.parity_slabs(33) == [0, 32, 31]  # first steps of both chunks, then the last ones
    assert bench.parity_slabs(40) == [0, 32, 31, 39]


def test_counter_traffic_is_quoted_only_for_the_profiled_sources():
    """roofline.traffic comes from the round's committed counter profile (profiles/r03_summary.json,
    falling back to r02's) -- but only while the sha of the HIP sources matches the one the profile
    was taken on, and only for the profiled workload"""
    found = None
    for name in ("r03_summary.json", "r02_summary.json"):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            s = json.load(f)
        if s["kernel_source_sha"] == bench.kernel_source_sha():
            found = (name, s)
            break
    gb, src = bench.measured_traffic(s["cells_per_launch"])
    if found is not None:
        name, s = found
        assert src == f"profiles/{name}"
        assert abs(gb * 1e9 - s["hbm_traffic_bytes_per_launch"]) < 1e7
        assert 1.0 <= s["hbm_traffic_bytes_per_cell"] / 16.0 < 1.05  # no wasted re-reads
    else:
        assert (gb, src) == (None, None)
    assert bench.measured_traffic(12345) == (None, None)  # another workload: never quoted


def test_p_process_cpu_baseline_plumbing():
    nz, ny, nx, nt = 4, 16, 24, 3
    g = synthetic.make_grid(ny, nx, nz)
    kw = dict(seed=synthetic.SEED, mask3d=g["volcello"])
    T = synthetic.field_numpy((nt, nz, ny, nx), field_id=1, lo=-2.0, scale=34.0, **kw)
    S = synthetic.field_numpy((nt, nz, ny, nx), field_id=2, lo=30.0, scale=10.0, **kw)
    masso = o.calc_masso(o.calc_rho(T, S, o.pressure_from_depth(g["z_l"])), g["volcello"])
    line = bench.cpu_baseline_processes(g, nz, ny, nx, nt, masso, procs=2, reps=1, timeout=120)
    assert "error" not in line, line
    assert line["cores"] == 2 and line["kind"] == "port" and line["unit"] == "Mcells/s"
    assert line["masso_max_rel_err_vs_gpu"] == 0.0  # the workers replay the same fields exactly
    assert "cpu" in line and line["value"] > 0
