#!/usr/bin/env python3
"""bench.py -- Mcells/s of the fused Wright-EOS + steric pass on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (N=1): BASELINE.json configs[2], the roofline config -- OM4 0.25-degree synthetic
grid 1440x1080x75, 120 time steps, fp64, global steric -- with theta/S (2 x 112 GB) resident
in HBM before the timed region starts.  One *step* = one pass of the hot path over that
batch: reference state (K0 rho0, volo; masso0 = masso(t=0) of the K1 launch) + K1 over all 120 time steps + the stage-2
reduce + the area sum + [N>1: one RCCL all-reduce of nt+3 doubles] + the host epilogue
(D2H of masso(t), log, scale).  A cell is one (t,z,y,x) grid point, wet or dry.

N>1 (BASELINE.json configs[3], weak scaling): the (yh,xh) plane is tiled 1x2 / 2x2 / 2x4 over
the ranks and every rank holds 150*N time steps of its tile resident (N=8: the 1200-step record
on 360x540 tiles, 279.9 GB of the card's 288 GiB), walked in 5 time chunks: K1 per chunk and ONE
asynchronous RCCL all-reduce of the chunk's masso(t) (+ volo, masso0, sum(area) in the first)
overlapping the next chunk's kernel; value = all ranks' cells / max-over-ranks time.

Besides the contract fields the JSON line carries
  roofline     -- K1 (k_steric_global): 16 algorithmic bytes per cell x cells per launch /
                  mean launch duration (HIP events on the launch stream, inside the timed
                  region) against the 8 TB/s HBM3E peak;
  cpu_baseline -- the numpy oracle (op-for-op restatement of the reference's CPU path) timed
                  on this host, one thread, on a bounded sample of the same fields (rank 0,
                  N=1 only) -- a reported baseline, not the target;
  parity       -- max relative difference GPU vs oracle on that sample (masso per slab);
  config5_f32  -- BASELINE.json configs[4]: after the float64 extras the record is replaced by
                  float32 theta/S (112 GB) and the global variants, the one-pass decomposition and
                  the local eta pass are timed on it in the product's default modes (and the
                  other float32 modes beside them), with one oracle slab in numpy float32 as the
                  check.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from momlevel_amd import _lib, core, engine, hostio, parallel, synthetic  # noqa: E402

# BASELINE.json's metric string, verbatim
METRIC = "Mcells/s for fused Wright-EOS+steric at 1440\u00d71080\u00d775; % HBM roofline"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_CELL = 16    # theta 8 + S 8; vol0/p are amortised over the time loop
GRID = (75, 1080, 1440)
NT_PER_GPU = 120        # N=1: BASELINE.json configs[2]
NT_PER_GPU_TILED = 150  # N>1: configs[3] -- 150*N steps of a 1/N tile (N=8: the 1200-step record)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nt", type=int, default=0,
                    help="time steps per GPU in full-grid equivalents (default 120 at N=1, 150 at N>1)")
    ap.add_argument("--chunks", type=int, default=5,
                    help="N>1: time chunks the record is walked in (one all-reduce each)")
    ap.add_argument("--grid", default=None, help="nz,ny,nx (default 75,1080,1440)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target seconds of CPU-baseline work (0 disables it)")
    ap.add_argument("--no-extras", action="store_true", help="skip the local-variant timings")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the P-process CPU line (-1: min(16, cores); 0 disables)")
    ap.add_argument("--detail-file", default=None,
                    help="also write the long-form JSON (the BENCH_DETAIL line) to this file")
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1 only: a world of ONE rank on the RCCL backend and the step of --gpus 8 "
                         "(time chunks, one asynchronous exchange per chunk, finish()) with the "
                         "collective forced; timed beside the plain N=1 step, no extras")
    ap.add_argument("--input-dtype", choices=["f64", "f32"], default="f64",
                    help="storage type of theta/S (f32 = BASELINE.json configs[4]; the headline is f64)")
    return ap.parse_args()


def kernel_source_sha():
    """sha256 of what the timed kernels are built from (csrc/build.py TIMED_SOURCES: the HIP
    sources and their headers, momlevel_promote.hip, include/momlevel_hip.h, and the compiler
    flags) -- a committed counter profile is only quoted while it still describes them."""
    from momlevel_amd.csrc.build import source_sha

    return source_sha()


# (earlier rounds' summaries carry the sha of fewer files and can never match again)
PROFILE_SUMMARIES = ("r06_summary.json",)
VARIANT_SUMMARIES = ("r06_variants_summary.json", "r06_f32_variants_summary.json",
                     "r06_strat_variants_summary.json")


def measured_traffic(cells_per_launch):
    """HBM bytes per K1 launch from the committed rocprofv3 --pmc passes (profiles/), if they
    were taken on this workload AND on these kernel sources; bench.py itself cannot read
    hardware counters.  A stale profile (sources changed since) yields null, not an old number."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in PROFILE_SUMMARIES:
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                s = json.load(f)
            if (s.get("cells_per_launch") == cells_per_launch
                    and s.get("kernel_source_sha") == kernel_source_sha()):
                return round(s["hbm_traffic_bytes_per_launch"] / 1e9, 2), f"profiles/{name}"
        except (OSError, KeyError, ValueError):
            pass
    return None, None


# VALU issue peak of the chip, in lane-instructions per second: 256 CUs x 4 SIMDs x 16 lanes per
# clock (a wave64 instruction occupies its SIMD for 4 clocks) x 2.4 GHz (MI355X_MICROARCH.md).
# float64 instructions measured 4.5-4.7 clocks and v_rcp_f64 15.5 (profiles/r01_tune_ops_*.log), so a
# kernel of float64 arithmetic saturates its SIMDs at ~0.85 of this figure.
VALU_PEAK_LANE_INSTR_PER_S = 256 * 4 * 16 * 2.4e9


def strat_source_sha():
    from momlevel_amd.csrc.build import strat_source_sha as sha

    return sha()


def valu_profiles():
    """{bench key: VALU instructions per cell} from the round's committed SQ_INSTS_VALU passes
    (profiles/r05_*variants_summary.json, written by scripts/summarize_variants.py) -- quoted, like
    roofline.traffic, only while the sha of the kernel sources matches the profiled ones."""
    here = os.path.dirname(os.path.abspath(__file__))
    found, sources = {}, []
    for name in VARIANT_SUMMARIES:
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                summ = json.load(f)
            if "strat" in name:  # the stratification kernels live in a source file of their own
                if summ.get("strat_source_sha") != strat_source_sha():
                    continue
            elif summ.get("kernel_source_sha") != kernel_source_sha():
                continue
            prefix = "config5_f32." if "f32" in name else ""
            for k in summ["kernels"]:
                for key in k.get("bench_keys") or ([k["bench_key"]] if k.get("bench_key") else []):
                    if "valu_wave_instr_per_cell" in k:
                        found[prefix + key] = k["valu_wave_instr_per_cell"]
            sources.append(f"profiles/{name}")
        except (OSError, KeyError, ValueError):
            pass
    return found, sources


def measure_valu_probe(dev):
    """This box's float64 VALU issue rate, lane-instructions per second: mlx_valu_probe (nothing but
    independent v_fma_f64 chains), best of three launches of ~9 ms (16384 fmas per chain: 2.7e11
    lane-instructions -- long enough that launch ramp and tail do not show: at 128 fmas per chain the
    launch lasts 70 us and reads 15 % low)."""
    core.valu_probe(256, dev)
    torch.cuda.synchronize(dev)
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = core.valu_probe(16384, dev)
        e1.record()
        torch.cuda.synchronize(dev)
        best = max(best, n / (e0.elapsed_time(e1) * 1e-3))
    return best


def add_valu_roofline(line, f64_probe=None):
    """For every timed kernel the committed counter profile covers: VALU instructions per cell and
    the fraction of the chip's VALU issue peak that rate amounts to -- the roofline that bounds the
    float32 and one-pass kernels (HBM traffic 1.03-1.08 x algorithmic, far below the HBM peak).
    ``f64_probe``: the live float64 issue rate of this box (measure_valu_probe) -- the practical
    ceiling, as stream_read_probe is for the HBM roofline."""
    instr, sources = valu_profiles()
    for key, per_cell in instr.items():
        node = line
        for part in key.split("."):
            node = node.get(part) if isinstance(node, dict) else None
        if not isinstance(node, dict):
            continue
        rate = node.get("Mcells/s")
        if rate is None and key == "roofline":
            rate = node["achieved"] / node["algorithmic_bytes_per_cell"] * 1e3  # GB/s / B -> Mcells/s
        if rate is None:
            continue
        node["valu_instr_per_cell"] = per_cell
        node["frac_of_valu_peak"] = round(per_cell * rate * 1e6 / VALU_PEAK_LANE_INSTR_PER_S, 4)
        if f64_probe:
            node["frac_of_f64_fma_probe"] = round(per_cell * rate * 1e6 / f64_probe, 4)
    line["valu_roofline"] = {
        "peak_lane_instr_per_s": VALU_PEAK_LANE_INSTR_PER_S,
        "f64_fma_probe_lane_instr_per_s": f64_probe and round(f64_probe, -9),
        "f64_fma_probe": "mlx_valu_probe: independent v_fma_f64 chains, nothing else -- this box's "
                         "float64 issue ceiling (a v_rcp_f64 costs 3.3 such instructions, so a "
                         "kernel with one reciprocal per cell saturates a little below 1.0)",
        "definition": "valu_instr_per_cell (SQ_INSTS_VALU x 64 / cells, committed profile of these "
                      "kernel sources) x cells/s / (256 CU x 4 SIMD x 16 lanes x 2.4 GHz)",
        "sources": sources or None,
    }


def per_kernel_table(line):
    """{bench key: [ms, frac_of_8TBs, frac_of_matching_probe, frac_of_f64_fma_probe]} for every
    timed extra of the line: the float64 rows, and of the float32 record (config5_f32) the rows of
    the product's default modes (the other float32 modes stay in the BENCH_DETAIL line)."""
    table = {}

    def walk(node, path):
        if not isinstance(node, dict):
            return
        if "ms" in node and "frac_of_8TBs" in node:
            table[".".join(path)] = [node["ms"], node["frac_of_8TBs"],
                                     node.get("frac_of_matching_probe"),
                                     node.get("frac_of_f64_fma_probe")]
            return
        for k, v in node.items():
            if path == ["config5_f32"] and k not in ("default",):
                continue
            if k in ("roofline", "config", "cpu_baseline", "valu_roofline"):
                continue
            walk(v, path + [k])

    walk(line, [])
    return {k.replace("config5_f32.default.", "f32."): v for k, v in table.items()}


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                 "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                 "cpu_baseline")


def compact_line(line, detail_bytes):
    """The contract line: the contract's keys, roofline (with the per-kernel table and the probes)
    and cpu_baseline whole; of everything else one line's worth."""
    out = {k: line[k] for k in CONTRACT_KEYS if k in line}
    for k in ("cpu_baseline_processes", "cpu_baseline_fused_openmp"):
        v = line.get(k)
        if isinstance(v, dict):
            out[k] = {kk: v[kk] for kk in ("value", "unit", "cores", "error") if kk in v}
    out["parity"] = line.get("parity")
    out["eta_t0_is_zero"] = line.get("eta_t0_is_zero")
    checks = {}

    def walk(node, path):
        if isinstance(node, dict):
            for k, v in node.items():
                if isinstance(v, bool) and ("bit_identical" in k or k.endswith("_is_zero")
                                            or "equals" in k or k.startswith("k1_default_is")):
                    checks[".".join(path + [k])] = v
                else:
                    walk(v, path + [k])

    walk({k: v for k, v in line.items() if k not in CONTRACT_KEYS}, [])
    if checks:
        out["checks_all_true"] = all(checks.values())
        out["checks_failed"] = [k for k, v in checks.items() if not v]
        out["checks_count"] = len(checks)
    if "forced_collective" in line:
        out["forced_collective"] = line["forced_collective"]
    ex = line.get("reference_example_call")
    if isinstance(ex, dict):
        out["reference_example_call"] = {k: ex[k] for k in (
            "wall_s", "GB/s_host_link_in_plus_out", "link_duplex_floor_s", "frac_of_link_duplex_floor",
            "step_bit_identical_to_oracle", "error") if k in ex}
    if "valu_roofline" in line:
        out["f64_fma_probe_lane_instr_per_s"] = line["valu_roofline"].get(
            "f64_fma_probe_lane_instr_per_s")
    out["detail"] = (f"the line before this one (prefix BENCH_DETAIL, {detail_bytes} bytes) holds "
                     "every row in full: kernel names, notes, checks, the non-default float32 modes")
    return out


def workload_config(world, grid, nt, nt_req, tile_hw, n_launches, chunk_steps, input_dtype,
                    backend, forced=False):
    """The ``config`` object of the JSON line: which BASELINE.json configuration this run IS (and
    says so only when it is: the record not shortened, the grid the 0.25-degree one, 8 ranks x 1200
    steps for configs[3]), the tile layout and the collective.  Pure: tests/test_bench_helpers.py
    checks the 8-rank 2x4 branch, which no one-GPU box can run."""
    nz, ny, nx = grid
    th, tw = tile_hw
    f32 = input_dtype == "f32"
    shrunk = nt < nt_req
    layout = {1: "1x1", 2: "1x2", 4: "2x2", 8: "2x4"}.get(world, f"1x{world}")
    cells_rank = nt * nz * th * tw
    if world == 1:
        workload = (
            f"OM4 0.25deg synthetic grid {nx}x{ny}x{nz}, {nt} time steps, "
            f"{'fp32 theta/S (BASELINE.json configs[4])' if f32 else 'fp64'}, global "
            + ("steric (shortened to fit free HBM: NOT BASELINE.json configs[2], see nt_requested)"
               if shrunk else "steric (BASELINE.json configs[2])")
            + (f"; walked as `--gpus 8` walks its tile: {n_launches} time chunks of <= "
               f"{chunk_steps} steps, one exchange per chunk FORCED through the backend in a world "
               "of one rank (--force-collective)" if forced else ""))
    else:
        workload = (
            f"OM4 0.25deg synthetic grid {nx}x{ny}x{nz} tiled {layout} (yh x xh), {nt} time "
            f"steps, fp64, global steric: every GPU holds all {nt} steps of its "
            f"{tw}x{th} tile resident = the cells of {nt / world:g} full-grid steps, walked "
            f"in {n_launches} time chunks of <= {chunk_steps} steps with one exchange "
            "per chunk ("
            + ("NOT BASELINE.json configs[3]: the record was shortened to fit free HBM, see "
               "nt_requested; " if shrunk else
               "BASELINE.json configs[3]; " if (world == 8 and nt == 1200
                                                and (nz, ny, nx) == GRID) else
               f"configs[3]'s tiling at {world} GPUs -- configs[3] itself is 8 GPUs x 1200 "
               "steps; ")
            + "weak scaling: bytes per GPU are fixed, the record grows with N)")
    return {
        "workload": workload,
        "grid_xyz": [nx, ny, nz],
        "nt_per_gpu_resident": nt,
        "nt_total": nt,
        "nt_requested": nt_req,
        "record_shortened_to_fit_hbm": shrunk,
        "tile_layout_yx": layout,
        "tile_xy": [tw, th],
        "variant": "steric",
        "domain": "global",
        "forced_collective": forced,
        "collective": ("none" if world == 1 and not forced else
                       f"{n_launches} per step, one per time chunk: {chunk_steps}(+3 in the first) "
                       f"f64 per rank, {parallel.exchange_mode()} "
                       + ("(all_gather_into_tensor + rank-ordered float64 sum: every element of "
                          "the vector is reduced in the same order)"
                          if parallel.exchange_mode() == "ordered" else "(all_reduce SUM)")
                       + ", asynchronous, overlapped with the next chunk's kernel"),
        "backend": (None if world == 1 and not forced else
                    "nccl (RCCL)" if backend == "nccl" else
                    backend + " (REHEARSAL: the ranks share GPUs and the exchange "
                    "is staged through the host; not an xGMI measurement)"),
        "time_chunks": n_launches,
        "input_dtype": input_dtype,
        "hbm_resident_gb": round(2 * cells_rank * (4 if f32 else 8) / 1e9, 1),
    }


def fit_nt(nt, nz, ny, nx, device, itemsize=8):
    """Largest nt <= requested whose theta+S fit in free HBM with ~14 GB of headroom."""
    free, _ = torch.cuda.mem_get_info(device)
    n3 = nz * ny * nx
    headroom = 14 * (1 << 30) + 6 * n3 * 8
    cap = int((free - headroom) // (2 * n3 * itemsize))
    return max(1, min(nt, cap))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def parity_slabs(nt, t_chunk=32):
    """Time steps whose GPU masso is checked against the oracle: first and last step of EVERY
    K1 time chunk that the bounded CPU budget allows, in an order that reaches all chunks first
    (t = 0, 32, 64, 96, then 31, 63, 95, 119 for nt=120)."""
    firsts = list(range(0, nt, t_chunk))
    lasts = [min(t + t_chunk, nt) - 1 for t in firsts]
    order = []
    for t in firsts + lasts:
        if t not in order:
            order.append(t)
    return order


def cpu_baseline(T, S, g, pres, target_s, gpu_masso):
    """Oracle timed on host cores over whole time slabs of the resident fields."""
    from oracle import momlevel_numpy as o  # the checker / timed baseline, never the product

    nz, ny, nx = T.shape[1:]
    cells = nz * ny * nx
    vol = g["volcello"]
    spent, slabs, err, done = 0.0, 0, 0.0, []
    for t in parity_slabs(T.shape[0])[:8]:
        Tn = hostio.to_host(T[t])  # (page-locked download: no GPU mapping of malloc'ed memory)
        Sn = hostio.to_host(S[t])
        t0 = time.perf_counter()
        rho = o.calc_rho(Tn, Sn, pres)
        m = o.calc_masso(rho, vol)
        spent += time.perf_counter() - t0
        slabs += 1
        done.append(t)
        err = max(err, abs(m - gpu_masso[t]) / abs(m))
        del rho, Tn, Sn
        if spent >= target_s and slabs >= 4:
            break
    return {
        "value": round(slabs * cells / spent / 1e6, 3),
        "unit": "Mcells/s",
        "cores": 1,
        "kind": "port",
        "cpu": cpu_model(),
        "sample": (f"{slabs} of {T.shape[0]} time slabs (t = {done}; {nx}x{ny}x{nz} cells each) of "
                   "the same synthetic fields: unfused numpy wright density + "
                   f"nansum(rho*volcello_ref), {spent:.1f} s on 1 thread of {cpu_model()}"),
    }, {"masso_max_rel_err_vs_oracle": float(err), "slabs_checked": slabs,
        "time_steps_checked": done,
        "note": "first/last step of K1 time chunks (32 steps each): the whole launch is covered"}


def cpu_baseline_processes(g, nz, ny, nx, nt, gpu_masso, procs, reps=2, timeout=240):
    """The P-process line of BASELINE.md section 4: P independent oracle processes, one time slab
    each, all timed together (how dask's chunks={"time": 1} spreads momlevel over a host).  Every
    worker regenerates its slab in numpy (no GPU, no shared memory) -- oracle/cpu_worker.py."""
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    steps = [int(round(i * (nt - 1) / max(1, procs - 1))) for i in range(procs)]
    workers = []
    try:
        for t in steps:
            workers.append(subprocess.Popen(
                [sys.executable, "-m", "oracle.cpu_worker", str(ny), str(nx), str(nz), str(t),
                 str(reps)], cwd=here, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                text=True))
        deadline = time.time() + timeout

        def line_from(w):
            """one stdout line of a worker, or a RuntimeError once the overall deadline has passed
            (a worker that died or hangs must never hang the bench)"""
            import select

            while True:
                left = deadline - time.time()
                if left <= 0:
                    raise RuntimeError("CPU worker timed out")
                ready, _, _ = select.select([w.stdout], [], [], min(left, 5.0))
                if ready:
                    text = w.stdout.readline()
                    if text == "":
                        raise RuntimeError(f"CPU worker exited with code {w.poll()}")
                    return text
                if w.poll() is not None:
                    raise RuntimeError(f"CPU worker exited with code {w.poll()}")

        for w in workers:
            if line_from(w).strip() != "ready":
                raise RuntimeError("worker failed to start")
        t0 = time.perf_counter()
        for w in workers:
            w.stdin.write("go\n")
            w.stdin.flush()
        outs = [line_from(w).split() for w in workers]
        wall = time.perf_counter() - t0
        per_slab = [float(o_[0]) for o_ in outs]
        err = max(abs(float(o_[1]) - gpu_masso[t]) / abs(float(o_[1])) for o_, t in zip(outs, steps))
    except Exception as exc:  # reported, never fatal for the bench line
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        for w in workers:
            try:
                w.kill()
            except Exception:
                pass
    cells = nz * ny * nx * reps * procs
    return {"value": round(cells / wall / 1e6, 1), "unit": "Mcells/s", "cores": procs,
            "kind": "port", "cpu": cpu_model(),
            "sample": (f"{procs} processes x {reps} passes over one time slab each (t = {steps}), "
                       f"unfused numpy oracle, wall {wall:.1f} s; slowest worker "
                       f"{max(per_slab):.2f} s/slab, fastest {min(per_slab):.2f}"),
            "masso_max_rel_err_vs_gpu": float(err)}


def cpu_baseline_fused(T, S, g, pres, gpu_masso, slabs=2):
    """Informative second CPU line: the oracle's fused, OpenMP C restatement (oracle/wright_fused.c)
    -- what a well-written multithreaded CPU implementation of the same pass achieves."""
    try:
        from oracle import wright_c  # checker / timed baseline, never the product
    except Exception:
        return None
    wright_c.set_threads(min(16, os.cpu_count() or 1))  # the 1-GPU box's CPU share is 16 cores
    vol = g["volcello"]
    wright_c.masso_slab(hostio.to_host(T[0]), hostio.to_host(S[0]), vol, pres)  # warm the thread pool
    spent, err = 0.0, 0.0
    for t in range(min(slabs, T.shape[0])):
        Tn, Sn = hostio.to_host(T[t]), hostio.to_host(S[t])
        t0 = time.perf_counter()
        m = wright_c.masso_slab(Tn, Sn, vol, pres)
        spent += time.perf_counter() - t0
        err = max(err, abs(m - gpu_masso[t]) / abs(m))
    cells = int(np.prod(T.shape[1:])) * min(slabs, T.shape[0])
    return {"value": round(cells / spent / 1e6, 1), "unit": "Mcells/s",
            "cores": wright_c.num_threads(), "kind": "port",
            "sample": f"{min(slabs, T.shape[0])} time slabs, fused C restatement with OpenMP",
            "masso_max_rel_err_vs_gpu": float(err)}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher -- N fresh rank processes (what
        # torch.distributed.run would start), before this process makes any GPU call; rank 0's JSON
        # line is relayed, the exit code is the worst rank's.  (device_count() does not initialise
        # the GPU; with fewer GPUs than ranks the ranks rehearse over gloo and say so.)
        sys.exit(parallel.launch_local_ranks(
            a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
            visible_gpus=torch.cuda.device_count()))
    rank, world, local_rank = parallel.init_from_env()
    if world != a.gpus:
        a.gpus = world
    core.require_device()
    import torch.distributed as dist

    forced = a.force_collective
    if forced:
        if world != 1:
            raise SystemExit("--force-collective is the N=1 rehearsal of the tiled step")
        # a world of one on the RCCL backend: communicator, collective stream, work.wait() -- all
        # that a one-GPU box can execute of what `--gpus 8` runs
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        port = parallel.rank_environments(1)[0]["MASTER_PORT"]
        dist.init_process_group(backend="nccl", rank=0, world_size=1,
                                init_method=f"tcp://127.0.0.1:{port}")
        a.no_extras, a.cpu_seconds = True, 0.0

    dev_index = local_rank % torch.cuda.device_count()  # == local_rank on a full node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    host_staged = world > 1 and dist.get_backend() == "gloo"  # rehearsal on fewer GPUs than ranks

    def allreduce_scalar(value, op, dtype):
        t = torch.tensor([value], dtype=dtype, device="cpu" if host_staged else dev)
        dist.all_reduce(t, op=op)
        return t.item()

    nz, ny, nx = (tuple(int(v) for v in a.grid.split(",")) if a.grid else GRID)
    tile = synthetic.tile_bounds(ny, nx, rank, world)
    th, tw = tile[1] - tile[0], tile[3] - tile[2]
    g = synthetic.make_grid(ny, nx, nz, tile=tile)
    if a.nt <= 0:
        a.nt = NT_PER_GPU if world == 1 else NT_PER_GPU_TILED
    nt_req = a.nt * world
    f32 = a.input_dtype == "f32"
    tdtype = torch.float32 if f32 else torch.float64
    bytes_per_cell = BYTES_PER_CELL // 2 if f32 else BYTES_PER_CELL
    nt = fit_nt(nt_req, nz, th, tw, dev, itemsize=4 if f32 else 8)
    if world > 1:  # every rank must run the same number of steps
        nt = int(allreduce_scalar(nt, dist.ReduceOp.MIN, torch.int64))
    shrunk = nt < nt_req  # free HBM did not hold the requested record: said so in the JSON line

    vol0 = hostio.to_device(g["volcello"], dev)  # (through page-locked staging, like the product)
    area = hostio.to_device(g["areacello"], dev)
    pres = torch.from_numpy(np.asarray(g["z_l"]) * 1.0e4 + 101325.0).to(dev)
    shape = (nt, nz, th, tw)
    kw = dict(seed=synthetic.SEED, mask3d=vol0, global_hw=(ny, nx), origin=g["origin"], device=dev)
    T = core.synth_field(shape, tdtype, field_id=synthetic.FIELD_THETAO,
                         lo=synthetic.THETA_LO, scale=synthetic.THETA_SCALE, **kw)
    S = core.synth_field(shape, tdtype, field_id=synthetic.FIELD_SO,
                         lo=synthetic.SO_LO, scale=synthetic.SO_SCALE, **kw)
    torch.cuda.synchronize(dev)

    launch_ms = []
    chunk_steps = -(-nt // max(1, a.chunks))

    def step_tiled(timed, force=forced):
        """N>1 (BASELINE.json configs[3]): the rank's tile of the whole record, walked in time
        chunks -- K1 per chunk, ONE asynchronous all-reduce per chunk overlapping the next chunk's
        kernel ([volo, masso0, sum(area)] ride in the first), epilogue replicated on every rank."""
        core.eos_map(T[0], S[0], pres)  # the reference state's rho0, as at N=1
        evs = []
        res = parallel.steric_global_tile_streamed(
            (T, S), vol0, area, pres, variants=("steric",), steps=chunk_steps, skip_dry=False,
            events=evs, force_collective=force)
        if timed:
            launch_ms.append(evs)
        return res["steric"]

    def step(timed):
        """One pass of the hot path over the resident batch (what momlevel.steric(global) does)."""
        if world > 1 or forced:
            return step_tiled(timed)
        return step_plain(timed)

    def step_plain(timed):
        _rho0, volo, _ = engine.reference_state(T[0], S[0], vol0, pres, with_masso=False)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        # skip_dry=False: the headline loads every cell, wet or dry, as the metric defines a cell
        # (BASELINE.md section 2); the product default skips dry lines (see "land_skipping" below)
        masso = engine.global_masso(T, S, vol0, pres, events=ev, skip_dry=False)
        masso0 = masso[0]  # reference slab = step 0 of the record (as steric() does)
        asum = core.nansum(area)
        red = parallel.exchange_global(masso, volo, masso0, asum)
        out = parallel.finalize(*red)  # D2H + host epilogue (synchronises)
        if timed:
            launch_ms.append([ev])
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1 or forced:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_loop(fn):
        for _ in range(a.warmup):
            res = fn(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            res = fn(True)
        fence()
        return res, time.perf_counter() - t0

    beside = None
    if forced:
        # the same record through (i) the plain N=1 step (one launch, no process-group call) and
        # (ii) the chunked walk WITHOUT a collective, so that what RCCL adds can be read off
        out_plain, el_plain = timed_loop(step_plain)
        out_walk, el_walk = timed_loop(lambda timed: step_tiled(timed, force=False))
        launch_ms.clear()
        before = dict(parallel.exchange_stats)
    out, elapsed = timed_loop(step)
    if forced:
        st = parallel.exchange_stats
        n = (a.steps + a.warmup)
        beside = {
            "ms_per_step_plain_one_launch": round(el_plain / a.steps * 1e3, 3),
            "ms_per_step_chunked_no_collective": round(el_walk / a.steps * 1e3, 3),
            "ms_per_step_chunked_rccl": round(elapsed / a.steps * 1e3, 3),
            "rccl_over_plain": round(elapsed / el_plain, 4),
            "rccl_over_chunked_no_collective": round(elapsed / el_walk, 4),
            "collectives_run": st["collectives"] - before["collectives"],
            "collectives_expected": n * max(1, -(-nt // chunk_steps)),
            "collectives_on_device": st["on_device"] - before["on_device"],
            "last_collective": st["last"],
            "masso_bit_identical_to_plain_step": bool(
                np.array_equal(out["masso"], out_plain["masso"])
                and np.array_equal(out["masso"], out_walk["masso"])),
            "eta_bit_identical_to_plain_step": bool(np.array_equal(out["eta"], out_plain["eta"])),
        }
    if world > 1:
        elapsed = float(allreduce_scalar(elapsed, dist.ReduceOp.MAX, torch.float64))

    k1_kernel = _lib.last_kernel()  # the instantiation the timed region's last K1 call launched
    cells_rank = nt * nz * th * tw
    cells_job = cells_rank * world
    # K1 time of one pass over the record = the sum over its chunk launches (one launch at N=1)
    k1_ms = float(np.mean([sum(e0.elapsed_time(e1) for e0, e1 in evs) for evs in launch_ms]))
    n_launches = len(launch_ms[0])
    achieved = bytes_per_cell * cells_rank / (k1_ms * 1e-3) / 1e9

    traffic, traffic_src = (None, None) if f32 else measured_traffic(cells_rank)

    extras = {}
    if not a.no_extras and world == 1:
        extras = local_variant_timings(T, S, vol0, pres, g, dev)

    cpu, parity, cpu_fused, cpu_procs = None, None, None, None
    if world == 1 and a.cpu_seconds > 0:
        cpu, parity = cpu_baseline(T, S, g, pres.cpu().numpy(), a.cpu_seconds, out["masso"])
        if not f32:  # the C restatement and the numpy replay of the fields are float64
            cpu_fused = cpu_baseline_fused(T, S, g, pres.cpu().numpy(), out["masso"])
            procs = min(16, os.cpu_count() or 1) if a.cpu_procs < 0 else a.cpu_procs
            if procs > 0:
                cpu_procs = cpu_baseline_processes(g, nz, ny, nx, nt, out["masso"], procs)

    if not a.no_extras and world == 1 and not f32:
        # BASELINE.json configs[4] in the driver-timed line: the float64 record makes room for the
        # float32 one (outputs of the float64 run are host-side by now)
        del T, S
        extras["config5_f32"] = f32_timings(vol0, pres, g, dev, nt, kw)
        if (nz, ny, nx) == GRID:
            extras["reference_example_call"] = example_call()

    if rank == 0:
        line = {
            "metric": METRIC,
            "value": round(cells_job * a.steps / elapsed / 1e6, 1),
            "unit": "Mcells/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if not f32 else "f64 (float32 theta/S: polynomial in f32 as numpy does, rest f64)",
            "data": "synthetic",
            "config": workload_config(world, (nz, ny, nx), nt, nt_req, (th, tw), n_launches,
                                      chunk_steps, a.input_dtype,
                                      dist.get_backend() if (world > 1 or forced) else None,
                                      forced=forced),
            "roofline": {
                "kernel": k1_kernel,  # mlx_last_kernel() after the timed launches
                "kernel_template_arguments": "<element type, cells per pack, packs per thread, "
                                             "variant, dtype mode, generic, skip_dry, fma>",
                "arith": core.arith_default("k1", tdtype) + " (the product default for this dtype)",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_unit": "GB per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc)",
                "traffic_source": traffic_src,
                "launch_ms": round(k1_ms, 4),
                "launches_per_step": n_launches,
                "algorithmic_bytes_per_cell": bytes_per_cell,
                "algorithmic_gb_per_launch": round(bytes_per_cell * cells_rank / 1e9, 2),
                "cells_per_launch": cells_rank,
                "time_loop_steps_per_block": min(nt if world == 1 and not forced else chunk_steps,
                                                 core.K1_TCHUNK["steric"]),
                "kernel_source_sha": kernel_source_sha(),
            },
            "cpu_baseline": cpu,
            "cpu_baseline_processes": cpu_procs,
            "cpu_baseline_fused_openmp": cpu_fused,
            "parity": parity,
            "eta_t0_is_zero": bool(out["eta"][0] == 0.0),
        }
        line.update(extras)
        if beside is not None:
            line["forced_collective"] = beside
        add_valu_roofline(line, measure_valu_probe(dev) if world == 1 else None)
        if extras:
            probes = {"f64" if not f32 else "f32": extras.get("probes")}
            if "config5_f32" in extras:
                probes["f32"] = extras["config5_f32"].get("probes")
            headline_probe = (extras.get("probes") or {}).get("2r")
            if headline_probe:
                line["roofline"]["frac_of_matching_probe"] = round(achieved / headline_probe, 4)
            line["roofline"]["probes"] = probes
            line["roofline"]["probes_unit"] = (
                "GB/s of mlx_stream_probe_mix (no arithmetic, one tile per block) per read:write mix "
                "(<streamed fields in>r[<float64 streams out>w]) on this record")
            line["roofline"]["per_kernel"] = per_kernel_table(line)
            line["roofline"]["per_kernel_columns"] = [
                "ms", "frac_of_8TBs (algorithmic bytes)", "frac_of_matching_probe",
                "frac_of_f64_fma_probe (null: no committed VALU profile of these sources)"]
        # The long form first (every row with its kernel name, notes and checks), the contract
        # line LAST and short: the driver keeps an 8 KB tail of stdout and parses the last line --
        # round 4's single 25 KB line lost every float64 extra from the record.
        detail = json.dumps(line)
        if a.detail_file:
            with open(a.detail_file, "w") as f:
                f.write(detail + "\n")
        if extras:
            print("BENCH_DETAIL " + detail, flush=True)
        print(json.dumps(compact_line(line, len(detail))), flush=True)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


def stratification_timings(T, S, g, steps, B1):
    """derived.calc_n2's kernel (csrc/momlevel_strat.hip: alpha, beta and both vertical derivatives
    in one pass; 2 x B1 B read + 8 B written per cell) on `steps` time steps of the resident record,
    against the no-math probe of that mix, with a band of columns of one step checked against the
    numpy oracle (numpy.gradient + the EOS module, as the reference evaluates it)."""
    from oracle import momlevel_numpy as o  # the checker, never the product

    nt, nz, ny, nx = T.shape
    plane = ny * nx
    free, _ = torch.cuda.mem_get_info(T.device)
    steps = int(min(steps, max(0, free - (4 << 30)) // (nz * plane * 8)))  # the float64 result
    if steps < 2:
        return {"skipped": "no room for the result beside the resident record"}
    z = np.asarray(g["z_l"], dtype=np.float64)
    pz = torch.from_numpy(z * 1.0e4 + 101325.0).to(T.device)
    Tc, Sc = T[:steps].reshape(steps, nz, plane), S[:steps].reshape(steps, nz, plane)
    cells = steps * nz * plane
    res = {}

    def run():
        res["n2"] = core.stratification(Tc, Sc, pz, z)

    ms = _time(run, reps=2)
    r = _rate(ms, 2 * B1 + 8, cells, mix="2r1w", kernel=False)
    r["steps"] = steps
    r["kernel"] = ("k_stratification<double, 2, kF64, kWright, MLX_STRAT_N2>" if B1 == 8 else
                   "k_stratification<float, 2, kF32Faithful, kWright, MLX_STRAT_N2>")
    band = min(ny, 96)  # whole columns (the derivative runs along z) of `band` rows of one step
    t = steps // 2
    got = res["n2"][t].reshape(nz, ny, nx)[:, :band].cpu().numpy()
    Tn, Sn = T[t, :, :band].cpu().numpy(), S[t, :, :band].cpu().numpy()
    t0 = time.perf_counter()
    ref = o.calc_n2(Tn, Sn, z)
    cpu_s = time.perf_counter() - t0
    r["band_bit_identical_to_numpy"] = bool(np.array_equal(got, ref, equal_nan=True))
    r["numpy_oracle_1_thread_Mcells/s"] = round(ref.size / cpu_s / 1e6, 2)  # (on that band)
    del res["n2"]
    torch.cuda.empty_cache()  # (the callers size their next buffers from the driver's free memory)
    return r


def local_slab_check(o, Tn, Sn, rho0, g, pres, drho_gpu, eta_gpu):
    """steric.py:151-166 on ONE time slab in numpy (the oracle's functions): delta_rho =
    where(volcello_ref notnull, rho - rho0, NaN), eta = -1/rhozero * nansum(dz * delta_rho) masked by
    the surface cell -- against the GPU's delta_rho / eta of that step.  -> bit-identical?"""
    vol = g["volcello"]
    rho = o.calc_rho(Tn, Sn, pres)
    dref = np.where(~np.isnan(vol), rho - rho0, np.nan)
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    eref = np.where(~np.isnan(vol[0]), (-1.0 / 1035.0) * o.nansum(dz * dref, axis=0), np.nan)
    ok = bool(np.array_equal(eta_gpu, eref, equal_nan=True))
    if drho_gpu is not None:
        ok = ok and bool(np.array_equal(drho_gpu, dref, equal_nan=True))
    return ok


def _example_call_checker(host, g, drho, eta, wall_s):
    """CPU leg of the example call: the oracle (numpy, op for op; one thread) on ONE time step of
    thermosteric(ds) -- its time extrapolated to the call, its delta_rho / eta against the GPU's."""
    from oracle import momlevel_numpy as o  # the checker / CPU baseline, never the product

    nt = host["thetao"].shape[0]
    pn = o.pressure_from_depth(g["z_l"])
    rho0 = o.calc_rho(host["thetao"][0], host["so"][0], pn)
    t = nt // 2
    t0 = time.perf_counter()
    rho = o.calc_rho(host["thetao"][t], host["so"][0], pn)
    dref = np.where(~np.isnan(g["volcello"]), rho - rho0, np.nan)
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    eref = np.where(~np.isnan(g["volcello"][0]), (-1.0 / 1035.0) * o.nansum(dz * dref, axis=0), np.nan)
    cpu = time.perf_counter() - t0
    return {"oracle_one_step_s_1_thread": round(cpu, 3),
            "oracle_whole_call_extrapolated_s": round(cpu * nt, 1),
            "speedup_vs_oracle_1_thread": round(cpu * nt / wall_s, 1),
            "step_bit_identical_to_oracle": bool(np.array_equal(drho[t], dref, equal_nan=True)
                                                 and np.array_equal(eta[t], eref, equal_nan=True))}


def example_call():
    """Informative, PCIe-INCLUSIVE, never the bench value: the reference's one recorded real-size
    call -- examples/example.ipynb cells 3-6, `thermosteric(ds)` on time 60 x z_l 35 x yh 1080 x xh
    1440 float32 fields with the default domain="local" -- through the PRODUCT's public signature on
    numpy-backed (host) inputs of that shape: upload through the staging ring, K2 (thermosteric,
    float32, delta_rho), delta_rho and eta back into numpy arrays; one step checked bit for bit
    against the oracle, whose one-thread time for that step is extrapolated to the call."""
    import importlib.util

    try:
        spec = importlib.util.spec_from_file_location(
            "example_call", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts",
                                         "example_call.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        torch.cuda.empty_cache()
        out = mod.run(reps=4, checker=_example_call_checker)
        torch.cuda.empty_cache()
        # the link's own floor for this byte mix, both directions at once, no host work at all
        # (scripts/link_duplex_probe.py): what the wall time is to be held against -- not
        # bytes / 57 GB/s: beside the uploads the downloads run at ~52 GB/s
        spec = importlib.util.spec_from_file_location(
            "link_duplex_probe", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts",
                                              "link_duplex_probe.py"))
        probe = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(probe)
        floor = probe.run(out["host_bytes_streamed_in_GB"], out["host_bytes_out_GB"], reps=3)
        torch.cuda.empty_cache()
        out["link_duplex_floor_s"] = floor["floor_s_for_this_byte_mix"]
        out["link_d2h_GB/s_alone_and_beside_h2d"] = [max(floor["d2h_alone"]["d2h_GB/s"]),
                                                     max(floor["both_at_once"]["d2h_GB/s"])]
        out["frac_of_link_duplex_floor"] = round(out["link_duplex_floor_s"] / min(out["wall_s"]), 3)
        out["note"] = ("wall_s[0] is the process's first call (page-locked rings and result mappings "
                       "are made there); from the second call on the results land in pooled mappings")
        return out
    except Exception as exc:  # reported, never fatal for the bench line
        return {"error": f"{type(exc).__name__}: {exc}"}


def _time(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def _rate(ms, bytes_per_cell, cells, mix=None, kernel=True):
    """One timed pass as a result row.  ``mix``: the read:write mix of its streams ("2r", "1r",
    "2r1w", "1r1w": streamed fields in, float64 streams out) -- add_probe_fractions() prices the
    row against the no-arithmetic probe of THAT mix on this box."""
    r = {"Mcells/s": round(cells / ms / 1e3, 1), "ms": round(ms, 3),
         "algorithmic_bytes_per_cell": bytes_per_cell,
         "GB/s": round(bytes_per_cell * cells / ms / 1e6, 1),
         "frac_of_8TBs": round(bytes_per_cell * cells / ms / 1e6 / HBM_PEAK_GBS, 4)}
    if kernel:  # the K1 / K2 instantiation the timed call launched (mlx_last_kernel)
        r["kernel"] = _lib.last_kernel()
    if mix:
        r["probe_mix"] = mix
    return r


def measure_probes(T, S, dbuf=None, starts=(), steps=0):
    """This box's streaming ceilings, GB/s, for the read:write mixes of the timed kernels on the
    record they run on: mlx_stream_probe_mix (no arithmetic; one tile per block, the fastest shape
    of scripts/tune_probe.hip's sweep for each mix).  Read-only mixes: one launch over the whole
    resident record.  Mixes with the float64 output stream: the record in chunks of ``steps`` into
    the reused buffer ``dbuf``, as the kernels with delta_rho run."""
    B1 = T.element_size()
    cells = T.numel()
    n3 = int(np.prod(T.shape[1:]))
    out = {"dtype": "float64" if B1 == 8 else "float32"}
    # (best of 5: a ceiling must not sit below a kernel by its own scatter -- kernels are best of 2-3)
    ms = _time(lambda: core.stream_probe_mix(T, None, write=False), reps=5)
    out["1r"] = round(B1 * cells / ms / 1e6, 1)
    ms = _time(lambda: core.stream_probe_mix(T, S, write=False), reps=5)
    out["2r"] = round(2 * B1 * cells / ms / 1e6, 1)
    if dbuf is not None and len(starts):
        done = len(starts) * steps * n3
        for nin in (1, 2):
            def run():
                for t0 in starts:
                    core.stream_probe_mix(T[t0:t0 + steps], S[t0:t0 + steps] if nin == 2 else None,
                                          out=dbuf, write=True)

            ms = _time(run, reps=5)
            out[f"{nin}r1w"] = round((nin * B1 + 8) * done / ms / 1e6, 1)
        out["write_chunk_steps"] = steps
    return out


def add_probe_fractions(node, probes):
    """frac_of_matching_probe = the row's GB/s over the probe of its mix, for every row below
    ``node`` that names one"""
    if not isinstance(node, dict):
        return
    mix = node.get("probe_mix")
    if mix and probes.get(mix) and "GB/s" in node:
        node["frac_of_matching_probe"] = round(node["GB/s"] / probes[mix], 4)
    for v in node.values():
        add_probe_fractions(v, probes)


def local_variant_timings(T, S, vol0, pres, g, dev):
    """Informative: the other variants on the same resident fields (outside the timed region)."""
    nt, nz, ny, nx = T.shape
    cells = nt * nz * ny * nx
    out = {}
    B1 = T.element_size()  # bytes per cell of ONE streamed field: 8 (float64) or 4 (float32)

    # K1's arithmetic: the product default for this dtype (fused on float64, exact on float32;
    # core.arith_default) carries the plain key, the other policy its name as a suffix
    default = core.arith_default("k1", T.dtype)
    out["k1_default_arith"] = default

    def tag(arith):
        return "" if arith == default else "_" + arith

    for arith in ("fused", "exact"):
        ms = _time(lambda: core.steric_global_masso(T, S[0], vol0, pres, skip_dry=False,
                                                    arith=arith))
        out["thermosteric_global" + tag(arith)] = _rate(ms, B1, cells, "1r")
        ms = _time(lambda: core.steric_global_masso(T[0], S, vol0, pres, skip_dry=False,
                                                    arith=arith))
        out["halosteric_global" + tag(arith)] = _rate(ms, B1, cells, "1r")
    other = "exact" if default == "fused" else "fused"
    ms = _time(lambda: core.steric_global_masso(T, S, vol0, pres, skip_dry=False, arith=other))
    out["steric_global" + tag(other)] = _rate(ms, 2 * B1, cells, "2r")
    # BASELINE.json configs[4]: steric + thermosteric + halosteric (+ heat content) from ONE pass
    # over theta/S, against the sum of the three single-variant launches
    for arith in ("fused", "exact"):
        ms = _time(lambda: core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False,
                                                     arith=arith))
        r = _rate(ms, 2 * B1, cells, "2r")
        r["note"] = "all three variants + sum(theta*vol0) per step, theta/S read once"
        out["decomposition_one_pass" + tag(arith)] = r
    # the product default (MLX_FLAG_SKIP_DRY): theta/S of all-dry 16-byte packs are never loaded;
    # bit-identical results, fewer HBM bytes than the 16 B/cell the metric counts
    ms = _time(lambda: core.steric_global_masso(T, S, vol0, pres, skip_dry=True))
    dry = float(torch.isnan(vol0).double().mean().item())
    out["land_skipping"] = {
        "steric_global_Mcells/s": round(cells / ms / 1e3, 1),
        "dry_cell_fraction": round(dry, 4),
        "note": "same outputs bit for bit; not the headline: the metric counts dry cells as loaded",
    }
    rho0m = core.fold_mask(core.eos_map(T[0], S[0], pres), vol0)
    zi = hostio.to_device(g["z_i"], dev)
    dep = hostio.to_device(g["deptho"], dev)
    eta = torch.empty((nt, ny, nx), dtype=torch.float64, device=dev)
    ms = _time(lambda: core.steric_local(T, S, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=zi,
                                         deptho=dep, want_delta_rho=False, eta_out=eta,
                                         skip_dry=False))
    out["local_eta_only"] = _rate(ms, 2 * B1, cells, "2r")
    probes = None
    # K2 with delta_rho: the 8 B/cell output does not fit beside the record, so it is produced in
    # time chunks into one reused buffer.  Two chunk sizes: 16 steps (the K2 thread's own time
    # block: one occupancy round per launch, 7 launches) and the largest the free HBM holds
    # (few launches: ramp and tail paid less often) -- VERDICT r2 weak #5.
    free, _ = torch.cuda.mem_get_info(dev)
    big = int(min(nt, max(0, free - (6 << 30)) // (nz * ny * nx * 8)))
    # whole time blocks of every K2 instantiation (8 or 16 steps per thread at float64)
    big = big // 96 * 96 if big >= 96 else (big // 48 * 48 if big >= 48 else big // 16 * 16)
    if big >= 32:
        dbig = torch.empty((big, nz, ny, nx), dtype=torch.float64, device=dev)
        starts_b = range(0, nt - big + 1, big)
        done_b = len(starts_b) * big * nz * ny * nx

        def run_big():
            for t0 in starts_b:
                core.steric_local(T[t0:t0 + big], S[t0:t0 + big], rho0m, vol0[0], pres,
                                  -1.0 / 1035.0, z_i=zi, deptho=dep, delta_rho_out=dbig,
                                  eta_out=eta[t0:t0 + big], skip_dry=False)

        ms_b = _time(run_big, reps=2)
        r = _rate(ms_b, 2 * B1 + 8, done_b, "2r1w")
        r["delta_rho_chunk_steps"] = big
        r["launches"] = len(starts_b)
        out["local_with_delta_rho_large_chunks"] = r
        # The held-field instantiations of the local pass (what momlevel.thermosteric(ds) /
        # halosteric(ds) run with the default domain="local", steric.py:150-166): one streamed
        # field in, delta_rho out -- 8 B read + 8 B written per cell at float64 (+ the held slab and
        # rho0m once per level and time block of the thread: 16/NTI B) -- against the probe with THAT
        # mix (1 stream in, 1 out), and the eta-only forms against the one-stream read probe.
        out.update(local_held_timings(T, S, rho0m, vol0, pres, zi, dep, eta, dbig, starts_b, big,
                                      g, B1))
        # this box's no-arithmetic ceilings for every mix, on this record and these chunks
        probes = measure_probes(T, S, dbig, starts_b, big)
        del dbig
        torch.cuda.empty_cache()
    if probes is None:
        probes = measure_probes(T, S)
    # continuity with rounds 2-4: the skipna sum of the theta record (grid-stride, one pack in
    # flight per thread) -- a slower shape than the "1r" probe, kept as a second opinion
    ms = _time(lambda: core.nansum(T))
    probes["1r_nansum_kernel"] = round(B1 * cells / ms / 1e6, 1)
    out["probes"] = probes
    chunk = min(nt, 16)
    free, _ = torch.cuda.mem_get_info(dev)
    if free > chunk * nz * ny * nx * 8 + (2 << 30):
        drho = torch.empty((chunk, nz, ny, nx), dtype=torch.float64, device=dev)
        starts = range(0, nt - chunk + 1, chunk)
        done = len(starts) * chunk * nz * ny * nx

        def run(skip):
            for t0 in starts:
                core.steric_local(T[t0:t0 + chunk], S[t0:t0 + chunk], rho0m, vol0[0], pres,
                                  -1.0 / 1035.0, z_i=zi, deptho=dep, delta_rho_out=drho,
                                  eta_out=eta[t0:t0 + chunk], skip_dry=skip)

        ms = _time(lambda: run(False), reps=2)
        out["local_with_delta_rho"] = _rate(ms, 2 * B1 + 8, done, "2r1w")
        out["local_with_delta_rho"]["delta_rho_chunk_steps"] = chunk
        out["local_with_delta_rho"]["launches"] = len(starts)

        # K0 (derived.calc_rho: the pointwise EOS map, 16 B read + 8 B written per cell), in the
        # same 16-step chunks
        def run_k0():
            for t0 in starts:
                rho = core.eos_map(T[t0:t0 + chunk], S[t0:t0 + chunk], pres)
                del rho

        kms = _time(run_k0, reps=2)
        out["calc_rho_map"] = _rate(kms, 2 * B1 + 8, done, "2r1w", kernel=False)
        ms_skip = _time(lambda: run(True), reps=2)
        out["land_skipping"]["local_with_delta_rho_Mcells/s"] = round(done / ms_skip / 1e3, 1)
        del drho
        torch.cuda.empty_cache()
        # out-of-contract extra (kept green, not extended): calc_n2
        out["calc_n2"] = stratification_timings(T, S, g, min(nt, 2 * chunk), B1)
        drho = torch.empty((chunk, nz, ny, nx), dtype=torch.float64, device=dev)

        # all three local variants: three launches (24 + 16 + 16 B/cell) vs ONE pass of the
        # all-variants K2 (16 B read + 3 x 8 B written per cell)
        def run_held(Tv, Sv):
            for t0 in starts:
                core.steric_local(Tv[t0:t0 + chunk] if Tv.dim() == 4 else Tv,
                                  Sv[t0:t0 + chunk] if Sv.dim() == 4 else Sv, rho0m, vol0[0], pres,
                                  -1.0 / 1035.0, z_i=zi, deptho=dep, delta_rho_out=drho,
                                  eta_out=eta[t0:t0 + chunk], skip_dry=False)

        three = ms + _time(lambda: run_held(T, S[0]), reps=2) + _time(lambda: run_held(T[0], S), reps=2)
        del drho
        c3 = max(1, chunk // 2)
        free, _ = torch.cuda.mem_get_info(dev)
        if free > 3 * c3 * nz * ny * nx * 8 + 3 * nt * ny * nx * 8 + (2 << 30):
            d3 = torch.empty((3, c3, nz, ny, nx), dtype=torch.float64, device=dev)
            e3 = torch.empty((3, nt, ny, nx), dtype=torch.float64, device=dev)
            starts3 = range(0, nt - c3 + 1, c3)
            done3 = len(starts3) * c3 * nz * ny * nx

            def run3():
                for t0 in starts3:
                    core.steric_local_decomp(T[t0:t0 + c3], S[t0:t0 + c3], T[0], S[0], rho0m,
                                             vol0[0], pres, -1.0 / 1035.0, z_i=zi, deptho=dep,
                                             delta_rho_out=d3, eta_out=e3[:, t0:t0 + c3],
                                             skip_dry=False)

            ms3 = _time(run3, reps=2)
            r = _rate(ms3, 2 * B1 + 24, done3)
            r["three_single_variant_launches_ms"] = round(three * done3 / done, 3)
            r["one_pass_speedup"] = round(three * done3 / done / ms3, 3)
            r["note"] = "steric + thermosteric + halosteric delta_rho and eta, theta/S read once"
            out["local_decomposition_one_pass"] = r
            del d3, e3
    add_probe_fractions(out, probes)
    return out


def local_held_timings(T, S, rho0m, vol0, pres, zi, dep, eta, dbuf, starts, steps, g, B1,
                        with_steric=False):
    """K2's held-field instantiations (+ the steric one for float32 records, ``with_steric``) with
    and without delta_rho on the resident record, in chunks of ``steps`` into the reused float64
    buffer ``dbuf``, and one slab of each variant against the oracle (numpy on the arrays' own
    dtype).  Each row names the read:write mix whose probe bounds it (add_probe_fractions)."""
    from oracle import momlevel_numpy as o  # the checker

    nt, nz, ny, nx = T.shape
    n3 = nz * ny * nx
    done = len(starts) * steps * n3
    out = {}

    def ops(variant, t0, t1):
        Tv = T[0] if variant == "halosteric" else T[t0:t1]
        Sv = S[0] if variant == "thermosteric" else S[t0:t1]
        return Tv, Sv

    def run(variant, want):
        for t0 in starts:
            Tv, Sv = ops(variant, t0, t0 + steps)
            core.steric_local(Tv, Sv, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=zi, deptho=dep,
                              want_delta_rho=want, delta_rho_out=dbuf if want else None,
                              eta_out=eta[t0:t0 + steps], skip_dry=False)

    variants = (("steric",) if with_steric else ()) + ("thermosteric", "halosteric")
    pn = pres.cpu().numpy()
    T0n, S0n = hostio.to_host(T[0]), hostio.to_host(S[0])
    rho0n = o.calc_rho(T0n, S0n, pn)
    t_chk = starts[-1] + steps - 1  # last step of the last chunk: still in dbuf after the run
    for variant in variants:
        nin = 2 if variant == "steric" else 1
        for want in (True, False):
            if variant == "halosteric" and not want:
                continue
            ms = _time(lambda: run(variant, want), reps=2)
            r = _rate(ms, nin * B1 + (8 if want else 0), done, f"{nin}r" + ("1w" if want else ""))
            if want:  # one slab against the oracle: the last step the run left in the buffer
                Tn = T0n if variant == "halosteric" else hostio.to_host(T[t_chk])
                Sn = S0n if variant == "thermosteric" else hostio.to_host(S[t_chk])
                r["slab_bit_identical_to_oracle"] = local_slab_check(
                    o, Tn, Sn, rho0n, g, pn, hostio.to_host(dbuf[steps - 1]),
                    hostio.to_host(eta[t_chk]))
                r["time_step_checked"] = int(t_chk)
            r["chunk_steps"], r["launches"] = steps, len(starts)
            key = "local_" + ("" if variant == "steric" else variant + "_") + (
                "with_delta_rho" if want else "eta_only")
            out[key] = r
    return out


def f32_timings(vol0, pres, g, dev, nt, synth_kw):
    """BASELINE.json configs[4]: float32 theta/S (what MOM6 writes) at the roofline grid, resident;
    global variants, the one-pass decomposition (+ heat) and the local eta pass.  "faithful" =
    numpy's own mixed precision (float32 polynomial), bit-identical to the reference pointwise and
    the product default; "upcast" = float64 arithmetic on the float32 values; "fused" = upcast with
    MLX_FLAG_FMA.  Algorithmic bytes: 4 B per streamed field and cell."""
    from oracle import momlevel_numpy as o  # the checker: one slab in numpy float32

    torch.cuda.empty_cache()
    nz, ny, nx = vol0.shape
    shape = (nt, nz, ny, nx)
    T = core.synth_field(shape, torch.float32, field_id=synthetic.FIELD_THETAO,
                         lo=synthetic.THETA_LO, scale=synthetic.THETA_SCALE, **synth_kw)
    S = core.synth_field(shape, torch.float32, field_id=synthetic.FIELD_SO,
                         lo=synthetic.SO_LO, scale=synthetic.SO_SCALE, **synth_kw)
    cells = nt * nz * ny * nx

    def rate(ms, bpc, mix):
        return _rate(ms, bpc, cells, mix)

    modes = {"faithful_fused": dict(f32_mode="faithful", arith="fused"),
             "faithful": dict(f32_mode="faithful", arith="exact"),
             "upcast": dict(f32_mode="upcast", arith="exact"),
             "upcast_fused": dict(f32_mode="upcast", arith="fused")}
    out = {"note": ("float32 theta/S resident (%.0f GB), %d steps.  faithful = numpy's float32 "
                    "polynomial, exact float64 tail: bit-identical pointwise, the product default "
                    "of K0/K2 on float32 input; faithful_fused = the same float32 polynomial with "
                    "the float64 tail fused (a few ulp from numpy on float32 input): the product "
                    "default of the global sums; upcast = float64 arithmetic on the float32 values "
                    "(1e-7 from numpy's float32 polynomial)" % (2 * cells * 4 / 1e9, nt))}
    for mode, kw in modes.items():
        r = {}
        r["steric"] = rate(_time(lambda: core.steric_global_masso(T, S, vol0, pres, skip_dry=False,
                                                                  **kw)), 8, "2r")
        r["thermosteric"] = rate(_time(lambda: core.steric_global_masso(
            T, S[0], vol0, pres, skip_dry=False, **kw)), 4, "1r")
        r["halosteric"] = rate(_time(lambda: core.steric_global_masso(
            T[0], S, vol0, pres, skip_dry=False, **kw)), 4, "1r")
        r["one_pass"] = rate(_time(lambda: core.steric_global_decomp(
            T, S, T[0], S[0], vol0, pres, skip_dry=False, **kw)), 8, "2r")
        r["one_pass"]["three_launches_ms"] = round(
            r["steric"]["ms"] + r["thermosteric"]["ms"] + r["halosteric"]["ms"], 3)
        out[mode] = r
    rho0m = core.fold_mask(core.eos_map(T[0], S[0], pres), vol0)
    zi = hostio.to_device(g["z_i"], dev)
    dep = hostio.to_device(g["deptho"], dev)
    eta = torch.empty((nt, ny, nx), dtype=torch.float64, device=dev)
    for mode, kw in modes.items():
        out[mode]["local_eta_only"] = rate(_time(lambda: core.steric_local(
            T, S, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=zi, deptho=dep, want_delta_rho=False,
            eta_out=eta, skip_dry=False, **kw)), 8, "2r")
    # what a user gets without choosing anything: the global sums in the fused-tail policy, the
    # local pass exact
    out["default"] = dict({k: out["faithful_fused"][k] for k in ("steric", "thermosteric",
                                                                  "halosteric", "one_pass")},
                          local_eta_only=out["faithful"]["local_eta_only"])
    # The local pass WITH delta_rho on the float32 record -- the reference's one recorded real-size
    # call is momlevel.thermosteric(ds) on float32 thetao/so with the default domain="local"
    # (examples/example.ipynb cell 6; steric.py:150-166): K2 VAR 2 / MLX_DTYPE_F32 with the float64
    # delta_rho store, 4 B read + 8 B written per cell.  The float64 delta_rho of all steps fits
    # beside the 112 GB record: one launch per variant.
    free, _ = torch.cuda.mem_get_info(dev)
    steps_d = int(min(nt, max(0, free - (6 << 30)) // (nz * ny * nx * 8)))
    if steps_d >= 6:
        # whole time blocks of the float32 K2 instantiations where the record allows (6, 12 or 16
        # steps per thread: the full 120-step record ends on a half-filled block of the 16-step ones)
        steps_d = steps_d // 24 * 24 if steps_d >= 24 else steps_d // 6 * 6
        dbuf = torch.empty((steps_d, nz, ny, nx), dtype=torch.float64, device=dev)
        starts_d = range(0, nt - steps_d + 1, steps_d)
        out["default"].update(local_held_timings(
            T, S, rho0m, vol0, pres, zi, dep, eta, dbuf, starts_d, steps_d, g, 4, with_steric=True))
        probes = measure_probes(T, S, dbuf, starts_d, steps_d)
        del dbuf
        torch.cuda.empty_cache()
    else:
        probes = measure_probes(T, S)
    out["probes"] = probes
    out["default"]["calc_n2"] = stratification_timings(T, S, g, min(nt, 48), 4)
    # derived.calc_pdens on float32 fields (derived.py:477: a python-float pressure, so numpy keeps
    # the whole expression float32): the any-dtype map, 8 B read + 4 B written per cell
    np_steps = min(nt, 40)
    pd_cells = np_steps * nz * ny * nx
    Tp, Sp = T[:np_steps].reshape(-1), S[:np_steps].reshape(-1)
    ms = _time(lambda: core.eos_map_promote(Tp, Sp, 101325.0))
    out["default"]["calc_pdens_map"] = {
        "ms": round(ms, 3), "steps": np_steps, "Mcells/s": round(pd_cells / ms / 1e3, 1),
        "GB/s": round(12 * pd_cells / ms / 1e6, 1),
        "frac_of_8TBs": round(12 * pd_cells / ms / 1e6 / HBM_PEAK_GBS, 4),
        "algorithmic_bytes_per_cell": 12, "kernel": "k_eos_promote<float, float, Weak, 4>"}
    # the default calls (no mode arguments) are those, and one whole slab agrees with numpy
    # evaluated on the float32 arrays (= what momlevel computes on float32 input)
    t = nt // 2
    rows = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False)
    k1_default = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False,
                                           **modes["faithful_fused"])
    exact = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False,
                                      **modes["faithful"])
    Tn, Sn, T0n, S0n = (hostio.to_host(x) for x in (T[t], S[t], T[0], S[0]))
    pn = pres.cpu().numpy()
    errs = {}
    for name, row, (a_, b_) in zip(("steric", "thermosteric", "halosteric"), rows.cpu().numpy(),
                                   ((Tn, Sn), (Tn, S0n), (T0n, Sn))):
        ref = o.calc_masso(o.calc_rho(a_, b_, pn), g["volcello"])
        errs[name] = float(abs(row[t] - ref) / abs(ref))
    pd = hostio.to_host(core.eos_map_promote(T[t].reshape(-1), S[t].reshape(-1), 101325.0))
    pd_ref = o.wright_density(Tn, Sn, 101325.0).reshape(-1)  # float32 throughout
    pd_same = bool(pd.dtype == pd_ref.dtype == np.float32
                   and np.array_equal(pd, pd_ref, equal_nan=True))
    out["parity"] = {"calc_pdens_float32_slab_bit_identical_to_numpy": pd_same,
                     "k1_default_is_faithful_fused": bool(torch.equal(rows, k1_default)),
                     "faithful_fused_vs_faithful_exact_max_rel": float(
                         ((k1_default[:3] - exact[:3]).abs() / exact[:3].abs()).max().item()),
                     "time_step_checked": t,
                     "masso_rel_err_vs_numpy_float32_oracle": errs,
                     "masso0_equals_masso_t0": bool(rows[0, 0] == rows[1, 0] == rows[2, 0])}
    add_probe_fractions(out, probes)
    del T, S
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
