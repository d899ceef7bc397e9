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
    """roofline.traffic comes from the round's committed counter profile (profiles/r05_summary.json)
    -- but only while the sha of the kernel sources (HIP sources, headers, public header, compiler
    flags: csrc/build.py TIMED_SOURCES) matches the one the profile was taken on, and only for the
    profiled workload"""
    found, s = None, None
    for name in bench.PROFILE_SUMMARIES:
        if not os.path.exists(os.path.join(ROOT, "profiles", name)):
            continue
        with open(os.path.join(ROOT, "profiles", name)) as f:
            s = json.load(f)
        if s["kernel_source_sha"] == bench.kernel_source_sha():
            found = (name, s)
            break
    if found is not None:
        name, s = found
        gb, src = bench.measured_traffic(s["cells_per_launch"])
        assert src == f"profiles/{name}"
        assert abs(gb * 1e9 - s["hbm_traffic_bytes_per_launch"]) < 1e7
        assert 1.0 <= s["hbm_traffic_bytes_per_cell"] / 16.0 < 1.05  # no wasted re-reads
    elif s is not None:
        assert bench.measured_traffic(s["cells_per_launch"]) == (None, None)  # stale: never quoted
    assert bench.measured_traffic(12345) == (None, None)  # another workload: never quoted


def test_kernel_source_sha_covers_headers_and_flags(monkeypatch):
    """the sha that guards the quoted counter profiles also covers the public
    header, momlevel_promote.hip (calc_pdens_map is timed from it) and the compiler flags"""
    from momlevel_amd.csrc import build

    names = {os.path.basename(p_) for p_ in build.TIMED_SOURCES}
    assert {"momlevel_hip.hip", "eos_device.hpp", "mlx_internal.hpp", "momlevel_promote.hip",
            "eos_promote.hpp", "momlevel_hip.h"} <= names
    before = bench.kernel_source_sha()
    assert before == build.source_sha() and len(before) == 16
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DSOMETHING"])
    assert bench.kernel_source_sha() != before


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
        "parallel.host_barrier(); dist.destroy_process_group()\n"
        "sys.exit(3 if (rank == 1 and os.environ.get('FAIL_RANK_1')) else 0)\n")
    out = io.StringIO()
    env = dict(os.environ, HIP_VISIBLE_DEVICES="")  # CPU ranks wherever the suite runs
    rc = parallel.launch_local_ranks(2, [sys.executable, "-c", prog], environ=env, visible_gpus=0,
                                     out=out)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 2, "sum": 3.0, "backend": "gloo"}
    assert "noise from rank 1" not in out.getvalue()  # only rank 0's stdout is relayed
    # the worst rank's exit code is the launcher's
    env = dict(os.environ, FAIL_RANK_1="1", HIP_VISIBLE_DEVICES="")
    assert parallel.launch_local_ranks(2, [sys.executable, "-c", prog], environ=env,
                                       visible_gpus=0, out=io.StringIO()) == 3


def test_self_launcher_at_eight_ranks():
    """the 8-rank launch of BASELINE.json configs[3] (`python bench.py --gpus 8`): 8 children, all
    reaped, rank 0 relayed, the worst exit code propagated -- rehearsed on the CPU over gloo, since
    a one-GPU box admits at most 6 processes on its card and the 8-GPU node is the driver's"""
    import io
    import sys

    from momlevel_amd import parallel

    prog = (
        "import json, os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from momlevel_amd import parallel, synthetic\n"
        "rank, world, local = parallel.init_from_env()\n"
        "tile = synthetic.tile_bounds(1080, 1440, rank, world)\n"
        "area = torch.tensor([float((tile[1]-tile[0])*(tile[3]-tile[2]))]); dist.all_reduce(area)\n"
        "print(json.dumps({'n_gpus': world, 'cells': area.item(), 'tile0': tile, "
        "'pid': os.getpid()}), flush=True) if rank == 0 else None\n"
        "parallel.host_barrier(); dist.destroy_process_group()\n"
        "sys.exit(5 if (rank == 6 and os.environ.get('FAIL_RANK_6')) else 0)\n")
    out = io.StringIO()
    env = dict(os.environ, HIP_VISIBLE_DEVICES="")
    rc = parallel.launch_local_ranks(8, [sys.executable, "-c", prog], environ=env, visible_gpus=0,
                                     out=out)
    assert rc == 0
    lines = [json.loads(l) for l in out.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8
    assert lines[0]["cells"] == 1080.0 * 1440.0 and lines[0]["tile0"] == [0, 540, 0, 360]
    env = dict(os.environ, FAIL_RANK_6="1", HIP_VISIBLE_DEVICES="")
    assert parallel.launch_local_ranks(8, [sys.executable, "-c", prog], environ=env,
                                       visible_gpus=0, out=io.StringIO()) == 5
    envs = parallel.rank_environments(8, environ={}, visible_gpus=8)
    assert [e["LOCAL_RANK"] for e in envs] == [str(r) for r in range(8)]
    assert all("MOMLEVEL_AMD_DIST_BACKEND" not in e for e in envs)  # a full node: RCCL


def test_workload_config_names_configs3_only_when_it_is():
    """the label branches no one-GPU box can reach: 8 ranks x 1200 steps on the 0.25-degree grid IS
    BASELINE.json configs[3] (2x4 tiles of 360 x 540); anything else says what it is instead"""
    c = bench.workload_config(8, (75, 1080, 1440), 1200, 1200, (540, 360), 5, 240, "f64", "nccl")
    assert c["tile_layout_yx"] == "2x4" and c["tile_xy"] == [360, 540]
    assert "BASELINE.json configs[3];" in c["workload"] and "NOT" not in c["workload"]
    assert c["backend"] == "nccl (RCCL)" and c["time_chunks"] == 5 and c["hbm_resident_gb"] == 279.9
    assert c["record_shortened_to_fit_hbm"] is False and "ordered" in c["collective"]
    short = bench.workload_config(8, (75, 1080, 1440), 1100, 1200, (540, 360), 5, 220, "f64", "nccl")
    assert "NOT BASELINE.json configs[3]" in short["workload"] and short["record_shortened_to_fit_hbm"]
    four = bench.workload_config(4, (75, 1080, 1440), 600, 600, (540, 720), 5, 120, "f64", "nccl")
    assert four["tile_layout_yx"] == "2x2" and "configs[3]'s tiling at 4 GPUs" in four["workload"]
    reh = bench.workload_config(8, (75, 1080, 1440), 16, 16, (540, 360), 2, 8, "f64", "gloo")
    assert "REHEARSAL" in reh["backend"] and "configs[3]'s tiling at 8 GPUs" in reh["workload"]
    one = bench.workload_config(1, (75, 1080, 1440), 120, 120, (1080, 1440), 1, 24, "f64", None)
    assert one["workload"].endswith("steric (BASELINE.json configs[2])") and one["collective"] == "none"
    assert one["tile_layout_yx"] == "1x1" and one["hbm_resident_gb"] == 223.9


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
    line = {"value": 1.0,
            "roofline": {"achieved": 6400.0, "algorithmic_bytes_per_cell": 16},
            "thermosteric_global": {"Mcells/s": 800000.0},
            "config5_f32": {"default": {"local_thermosteric_with_delta_rho": {"Mcells/s": 380000.0}},
                            "faithful_fused": {"one_pass": {"Mcells/s": 400000.0}}}}
    bench.add_valu_roofline(line, f64_probe=30.0e12)
    assert line["valu_roofline"]["peak_lane_instr_per_s"] == 256 * 4 * 16 * 2.4e9
    name = bench.VARIANT_SUMMARIES[0]  # the round's float64 variants profile
    path = os.path.join(ROOT, "profiles", name)
    current = False
    if os.path.exists(path):
        with open(path) as f:
            current = json.load(f)["kernel_source_sha"] == bench.kernel_source_sha()
    if not current:  # no profile of THESE sources: nothing is quoted
        assert not any(k for k in instr if "calc_n2" not in k)
        assert "valu_instr_per_cell" not in line["roofline"]
        return
    assert f"profiles/{name}" in sources
    # round 6: the two timings of the K2-with-delta_rho instantiation share one profile row, and K0's
    # map has its row -- no per_kernel row of the float64 table is left without a VALU column
    assert instr["local_with_delta_rho"] == instr["local_with_delta_rho_large_chunks"]
    assert "calc_rho_map" in instr and "config5_f32.default.calc_pdens_map" in instr
    r = line["roofline"]  # 6400 GB/s at 16 B/cell = 400 Gcells/s
    assert r["valu_instr_per_cell"] == instr["roofline"]
    assert abs(r["frac_of_valu_peak"] - instr["roofline"] * 400e9 / (256 * 4 * 16 * 2.4e9)) < 1e-3
    assert abs(r["frac_of_f64_fma_probe"] - instr["roofline"] * 400e9 / 30.0e12) < 1e-3
    t = line["thermosteric_global"]
    assert t["valu_instr_per_cell"] == instr["thermosteric_global"] < instr["thermosteric_global_exact"]


def test_the_contract_line_is_short_and_carries_every_kernel():
    """the driver keeps 8 KB of stdout.  The last line is the contract line --
    contract keys, roofline (+ per_kernel table, probes), cpu_baseline -- and stays well under
    that; the long form travels as an earlier BENCH_DETAIL line."""
    with open(os.path.join(ROOT, "profiles", "r04_bench_line.json")) as f:
        line = json.load(f)
    line["roofline"]["per_kernel"] = bench.per_kernel_table(line)
    table = line["roofline"]["per_kernel"]
    for key in ("local_with_delta_rho", "local_eta_only", "local_thermosteric_with_delta_rho",
                "local_thermosteric_eta_only", "thermosteric_global", "calc_rho_map",
                "f32.local_thermosteric_with_delta_rho", "f32.one_pass", "f32.calc_pdens_map"):
        assert key in table and len(table[key]) == 4, key
    assert not any(k.startswith(("config5_f32", "f32.upcast", "f32.faithful")) for k in table)
    assert table["local_thermosteric_eta_only"][0] == line["local_thermosteric_eta_only"]["ms"]
    assert len(json.dumps(table)) < 2400
    short = bench.compact_line(line, 25000)
    for k in bench.CONTRACT_KEYS:
        assert short[k] == line[k]
    assert len(json.dumps(short)) < 6500
    assert short["checks_all_true"] is True and short["checks_count"] >= 10


def test_probe_fractions_follow_the_rows_mix():
    node = {"a": {"GB/s": 5000.0, "probe_mix": "1r1w"}, "b": {"c": {"GB/s": 3000.0, "probe_mix": "2r"}},
            "d": {"GB/s": 1.0, "probe_mix": "9r"}, "e": {"GB/s": 1.0}}
    bench.add_probe_fractions(node, {"1r1w": 6250.0, "2r": 6000.0, "dtype": "float64"})
    assert node["a"]["frac_of_matching_probe"] == 0.8
    assert node["b"]["c"]["frac_of_matching_probe"] == 0.5
    assert "frac_of_matching_probe" not in node["d"] and "frac_of_matching_probe" not in node["e"]
