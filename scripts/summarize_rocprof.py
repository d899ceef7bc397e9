"""Condense rocprofv3 CSV output (gpurun_out/prof_*/) into the small summaries kept under profiles/.

    python scripts/summarize_rocprof.py gpurun_out/prof_r01 profiles/r01

Writes <prefix>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
<prefix>_k1_dispatches.csv (every k_steric_global dispatch: grid, duration) and
<prefix>_summary.json (mean duration of the full-batch K1 launches; HBM traffic from the
FETCH_SIZE / WRITE_SIZE passes with the gfx950 corrections of MI355X_MICROARCH.md section HBM:
counters are in KiB; FETCH_SIZE reads exactly 1/2 of a wide coalesced stream -> x2).
"""

import csv
import glob
import json
import os
import shutil
import sys


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def main(src, prefix, kernel="k_steric_global"):
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    summary = {"kernel": kernel}
    stats = one(os.path.join(src, "trace", "**", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats, prefix + "_kernel_stats.csv")
    trace = one(os.path.join(src, "trace", "**", "*_kernel_trace.csv"))
    if trace:
        rows = [r for r in csv.DictReader(open(trace)) if kernel in r["Kernel_Name"]]
        with open(prefix + "_k1_dispatches.csv", "w") as f:
            f.write("dispatch_id,kernel,grid_size,workgroup_size,vgpr,lds_bytes,duration_ns\n")
            for r in rows:
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                r["dur"] = dur
                grid = "x".join(r[f"Grid_Size_{a}"] for a in "XYZ")
                wg = "x".join(r[f"Workgroup_Size_{a}"] for a in "XYZ")
                f.write(f'{r["Dispatch_Id"]},"{r["Kernel_Name"][:60]}",{grid},'
                        f'{wg},{r["VGPR_Count"]},{r["LDS_Block_Size"]},{dur}\n')
        # the full-batch launches (nt = all resident steps) are the long ones; the short ones
        # are the nt=1 launches of the reference state (same grid, 1 time step)
        durs = sorted(r["dur"] for r in rows)
        cut = max(durs) / 2
        full = [d for d in durs if d > cut]
        ref = [d for d in durs if d <= cut]
        summary["full_batch_launches"] = len(full)
        summary["full_batch_mean_ms"] = sum(full) / len(full) / 1e6
        summary["full_batch_min_ms"] = min(full) / 1e6
        summary["full_batch_max_ms"] = max(full) / 1e6
        summary["reference_state_launches"] = len(ref)
        summary["reference_state_mean_ms"] = (sum(ref) / len(ref) / 1e6) if ref else None
        summary["grid_size_threads_xyz"] = [int(rows[0][f"Grid_Size_{a}"]) for a in "XYZ"]
    for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = one(os.path.join(src, tag, "**", "*_counter_collection.csv"))
        if not f:
            continue
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
        full = [v for v in vals if v > max(vals) / 2]
        summary[f"{counter}_KiB_per_full_launch"] = sum(full) / len(full)
        summary[f"{counter}_launches"] = len(full)
    if "FETCH_SIZE_KiB_per_full_launch" in summary:
        fetch = summary["FETCH_SIZE_KiB_per_full_launch"] * 1024 * 2  # gfx950: x2 (see docstring)
        write = summary.get("WRITE_SIZE_KiB_per_full_launch", 0.0) * 1024
        summary["hbm_read_bytes_per_launch_corrected"] = fetch
        summary["hbm_write_bytes_per_launch"] = write
        summary["hbm_traffic_bytes_per_launch"] = fetch + write
    # the bench line printed by the profiled command (same workload as the committed BENCH line)
    for log in ("bench_trace.log", "bench_pmc_fetch.log", "bench_pmc_write.log"):
        path = os.path.join(src, log)
        if os.path.exists(path):
            for line in open(path):
                if line.startswith("{") and '"roofline"' in line:
                    b = json.loads(line)
                    summary.setdefault("bench_lines", {})[log] = {
                        "value": b["value"], "ms_per_step": b["ms_per_step"],
                        "roofline": b["roofline"], "workload": b["config"]["workload"],
                    }
                    summary["cells_per_launch"] = b["roofline"]["cells_per_launch"]
    if "hbm_traffic_bytes_per_launch" in summary and "cells_per_launch" in summary:
        summary["hbm_traffic_bytes_per_cell"] = (
            summary["hbm_traffic_bytes_per_launch"] / summary["cells_per_launch"])
    # the sha bench.py compares against: a counter profile is quoted only for the sources it was
    # taken on (bench.py kernel_source_sha)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from momlevel_amd.csrc.build import source_sha

    summary["kernel_source_sha"] = source_sha()
    json.dump(summary, open(prefix + "_summary.json", "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
