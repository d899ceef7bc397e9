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
    """roofline.traffic comes from the round's committed counter profile (profiles/r04_summary.json,
    falling back to earlier rounds') -- but only while the sha of the HIP sources matches the one the profile
    was taken on, and only for the profiled workload"""
    found = None
    for name in ("r04_summary.json", "r03_summary.json", "r02_summary.json"):
        if not os.path.exists(os.path.join(ROOT, "profiles", name)):
            continue
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


def test_rank_environments_are_what_torchrun_would_set():
    from momlevel_amd import parallel

    envs = parallel.rank_environments(4, environ={"PATH": "/bin"}, port=29123, visible_gpus=8)
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    for e in envs:
        assert e["WORLD_SIZE"] == e["LOCAL_WORLD_SIZE"] == "4"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29123"
        assert e["PATH"] == "/bin" and "MOMLEVEL_AMD_DIST_BACKEND" not in e  # 8 GPUs: RCCL
    # fewer GPUs than ranks: the rehearsal backend, unless the caller chose one
    assert all(e["MOMLEVEL_AMD_DIST_BACKEND"] == "gloo"
               for e in parallel.rank_environments(2, environ={}, visible_gpus=1))
    assert all(e["MOMLEVEL_AMD_DIST_BACKEND"] == "nccl" for e in parallel.rank_environments(
        2, environ={"MOMLEVEL_AMD_DIST_BACKEND": "nccl"}, visible_gpus=1))
    # a free port is picked when none is given, the same for every rank
    ports = {e["MASTER_PORT"] for e in parallel.rank_environments(3, environ={})}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_self_launcher_runs_the_ranks_and_relays_rank_zero():
    """`python bench.py --gpus N` without torchrun: bench.py starts its own N rank processes
    (parallel.launch_local_ranks) before touching the GPU.  Here the ranks are a two-line program
    that does what bench.py's ranks do first -- parallel.init_from_env() -- and then all-reduces
    its rank over gloo; rank 0 prints one JSON line."""
    import io
    import sys

    from momlevel_amd import parallel

    prog = (
        "import json, os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from momlevel_amd import parallel\n"
        "rank, world, local = parallel.init_from_env()\n"
        "t = torch.tensor([float(rank + 1)]); dist.all_reduce(t)\n"
        "print('noise from rank', rank) if rank else print(json.dumps({'n_gpus': world, "
        "'sum': t.item(), 'backend': dist.get_backend()}), flush=True)\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "sys.exit(3 if (rank == 1 and os.environ.get('FAIL_RANK_1')) else 0)\n")
    out = io.StringIO()
    rc = parallel.launch_local_ranks(2, [sys.executable, "-c", prog], visible_gpus=0, out=out)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 2, "sum": 3.0, "backend": "gloo"}
    assert "noise from rank 1" not in out.getvalue()  # only rank 0's stdout is relayed
    # the worst rank's exit code is the launcher's
    env = dict(os.environ, FAIL_RANK_1="1")
    assert parallel.launch_local_ranks(2, [sys.executable, "-c", prog], environ=env,
                                       visible_gpus=0, out=io.StringIO()) == 3


def test_bench_main_becomes_the_launcher_before_any_gpu_call(monkeypatch):
    """--gpus 2 without WORLD_SIZE: main() hands over to the launcher with its own command line and
    never reaches require_device()"""
    import sys

    import pytest

    seen = {}

    def fake_launch(n, argv, visible_gpus=None, **kw):
        seen.update(n=n, argv=argv, visible_gpus=visible_gpus)
        return 0

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.parallel, "launch_local_ranks", fake_launch)
    monkeypatch.setattr(bench.core, "require_device",
                        lambda: (_ for _ in ()).throw(AssertionError("GPU touched")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--nt", "10", "--steps", "2"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0
    assert seen["n"] == 2 and seen["argv"][1].endswith("bench.py")
    assert seen["argv"][2:] == ["--gpus", "2", "--nt", "10", "--steps", "2"]


def test_valu_roofline_fields_come_from_the_profile_of_these_sources():
    """bench.add_valu_roofline: every timed kernel that the round's committed SQ_INSTS_VALU profile
    covers gets valu_instr_per_cell / frac_of_valu_peak / frac_of_f64_fma_probe -- quoted, like
    roofline.traffic, only while the profile's kernel-source sha is the current one"""
    instr, sources = bench.valu_profiles()
    with open(os.path.join(ROOT, "profiles", "r04_variants_summary.json")) as f:
        summ = json.load(f)
    line = {"value": 1.0,
            "roofline": {"achieved": 6400.0, "algorithmic_bytes_per_cell": 16},
            "thermosteric_global": {"Mcells/s": 800000.0},
            "config5_f32": {"default": {"local_thermosteric_with_delta_rho": {"Mcells/s": 380000.0}},
                            "faithful_fused": {"one_pass": {"Mcells/s": 400000.0}}}}
    bench.add_valu_roofline(line, f64_probe=30.0e12)
    assert line["valu_roofline"]["peak_lane_instr_per_s"] == 256 * 4 * 16 * 2.4e9
    if summ["kernel_source_sha"] != bench.kernel_source_sha():
        assert instr == {} and sources == [] and "valu_instr_per_cell" not in line["roofline"]
        return
    assert "profiles/r04_variants_summary.json" in sources
    r = line["roofline"]  # 6400 GB/s at 16 B/cell = 400 Gcells/s
    assert r["valu_instr_per_cell"] == instr["roofline"]
    assert abs(r["frac_of_valu_peak"] - instr["roofline"] * 400e9 / (256 * 4 * 16 * 2.4e9)) < 1e-3
    assert abs(r["frac_of_f64_fma_probe"] - instr["roofline"] * 400e9 / 30.0e12) < 1e-3
    t = line["thermosteric_global"]
    assert t["valu_instr_per_cell"] == instr["thermosteric_global"] < instr["thermosteric_global_exact"]
    k2 = line["config5_f32"]["default"]["local_thermosteric_with_delta_rho"]
    assert k2["valu_instr_per_cell"] == instr["config5_f32.default.local_thermosteric_with_delta_rho"]
    assert "frac_of_f64_fma_probe" in line["config5_f32"]["faithful_fused"]["one_pass"]
