"""Condense the rocprofv3 passes over scripts/profile_variants.py into profiles/<prefix>_variants_summary.json.

    python scripts/summarize_variants.py gpurun_out/prof_r02v profiles/r02

Expects under <src>/: plan.json (written by profile_variants.py --plan-out), plain.log (the
script's own event-timed JSON lines, unprofiled), and one rocprofv3 output directory per pass:
trace/ (--kernel-trace --stats), pmc_fetch/ (FETCH_SIZE), pmc_write/ (WRITE_SIZE), pmc_sq/ (SQ_*).
Dispatches are matched to cases by ORDER: the main kernels (k_steric_global / k_steric_local) of a
pass appear in the order and multiplicity of the plan.  HBM bytes = 2*FETCH_SIZE*1024 +
WRITE_SIZE*1024 (counters are KiB; gfx950 reports half of a wide coalesced read --
MI355X_MICROARCH.md, HBM).  SQ counters: totals over all waves of the dispatch.
"""

import csv
import glob
import json
import os
import shutil
import sys

# the kernels whose dispatches the plan lists (MLX_SUMMARY_MAIN overrides: run_profiles_strat.sh)
MAIN = tuple(os.environ.get("MLX_SUMMARY_MAIN",
                            "k_steric_global,k_steric_local,k_eos_map,k_eos_promote").split(","))


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd.csrc.build import source_sha as kernel_source_sha, strat_source_sha  # noqa: E402


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def by_dispatch(csv_path, value_cols):
    """main-kernel dispatches in order -> list of dicts {kernel_name, <cols>...}"""
    rows = {}
    for r in csv.DictReader(open(csv_path)):
        if not any(k in r["Kernel_Name"] for k in MAIN):
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        value_cols(r, d)
    return [rows[k] for k in sorted(rows)]


def chunk(dispatches, plan):
    out, i = [], 0
    for c in plan["cases"]:
        part = dispatches[i:i + c["launches"]]
        i += c["launches"]
        for d in part:
            assert c["kernel"] in d["name"], (c["case"], d["name"])
        out.append(part)
    assert i == len(dispatches), (i, len(dispatches))
    return out


def main(src, prefix):
    plan = json.load(open(os.path.join(src, "plan.json")))
    cells = plan["cells_per_launch"]
    summary = {
        "grid": "{}x{}x{}, nt={} resident, {}".format(*plan["grid"], plan["nt"], plan["dtype"]),
        "nt_of_cases_with_a_4d_output": plan.get("nt_of_cases_with_a_4d_output", plan["nt"]),
        # bench.py quotes these instruction counts only while the kernel sources are the profiled ones
        "kernel_source_sha": kernel_source_sha(),
        "strat_source_sha": strat_source_sha() if "k_stratification" in MAIN else None,
        "cells_per_launch": cells,
        "notes": __doc__.split("Expects")[1].strip().replace("\n", " "),
        "kernels": [],
    }
    plain = os.path.join(src, "plain.log")
    if os.path.exists(plain):
        summary["event_timed_unprofiled_run"] = [json.loads(l) for l in open(plain)
                                                 if l.startswith("{")]
    trace = one(os.path.join(src, "trace", "**", "*_kernel_trace.csv"))
    stats = one(os.path.join(src, "trace", "**", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats, prefix + "_variants_kernel_stats.csv")

    def dur(r, d):
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d["vgpr"], d["agpr"], d["lds"] = r["VGPR_Count"], r.get("Accum_VGPR_Count"), r["LDS_Block_Size"]
        d["grid"] = "x".join(r[f"Grid_Size_{a}"] for a in "XYZ")

    def counter(r, d):
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])

    tr = chunk(by_dispatch(trace, dur), plan) if trace else None
    passes = {}
    for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
        f = one(os.path.join(src, tag, "**", "*_counter_collection.csv"))
        if f:
            passes[tag] = chunk(by_dispatch(f, counter), plan)
    for i, c in enumerate(plan["cases"]):
        if c.get("setup"):  # (a dispatch of the set-up, listed so that the order matches; no row)
            continue
        cells = c.get("cells", plan["cells_per_launch"])  # (per case since round 6)
        keys = c.get("bench_key")
        k = {"kernel": c["case"], "algorithmic_bytes_per_cell": c["algorithmic_bytes_per_cell"],
             "cells_per_launch": cells,
             "bench_key": keys[0] if isinstance(keys, list) else keys,
             "bench_keys": keys if isinstance(keys, list) else ([keys] if keys else [])}
        if tr:
            part = tr[i][1:] or tr[i]  # first launch of a case is the warm-up
            ms = sum(d["ns"] for d in part) / len(part) / 1e6
            k.update(kernel_name=tr[i][0]["name"].split("(")[0].replace("void mlx::", ""),
                     vgpr=tr[i][0]["vgpr"], accum_vgpr=tr[i][0]["agpr"], lds_bytes=tr[i][0]["lds"],
                     grid_threads=tr[i][0]["grid"], rocprof_mean_ms=round(ms, 3),
                     Mcells_per_s=round(cells / ms / 1e3, 1),
                     algorithmic_GBps=round(c["algorithmic_bytes_per_cell"] * cells / ms / 1e6, 1),
                     frac_of_8TBps=round(c["algorithmic_bytes_per_cell"] * cells / ms / 1e6 / 8000, 4))
        rd = wr = None
        if "pmc_fetch" in passes:
            v = [d["FETCH_SIZE"] for d in passes["pmc_fetch"][i]]
            rd = 2 * 1024 * sum(v) / len(v)
            k["hbm_read_GB"] = round(rd / 1e9, 2)
        if "pmc_write" in passes:
            v = [d["WRITE_SIZE"] for d in passes["pmc_write"][i]]
            wr = 1024 * sum(v) / len(v)
            k["hbm_write_GB"] = round(wr / 1e9, 2)
        if rd is not None and wr is not None:
            k["hbm_bytes_per_cell"] = round((rd + wr) / cells, 3)
            k["traffic_over_algorithmic"] = round((rd + wr) / cells / c["algorithmic_bytes_per_cell"], 3)
            if tr:
                k["hbm_GBps_counters"] = round((rd + wr) / (k["rocprof_mean_ms"] * 1e-3) / 1e9, 1)
        if "pmc_sq" in passes:
            n = len(passes["pmc_sq"][i])
            sq = {name: sum(d.get(name, 0.0) for d in passes["pmc_sq"][i]) / n
                  for name in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
                               "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES")}
            k["valu_wave_instr_per_cell"] = round(sq["SQ_INSTS_VALU"] * 64 / cells, 2)
            wc = sq["SQ_WAVE_CYCLES"] or 1.0
            k["valu_active_over_wave_cycles"] = round(sq["SQ_ACTIVE_INST_VALU"] / wc, 3)
            k["wait_any_over_wave_cycles"] = round(sq["SQ_WAIT_ANY"] / wc, 3)
            k["wait_inst_any_over_wave_cycles"] = round(sq["SQ_WAIT_INST_ANY"] / wc, 3)
        summary["kernels"].append(k)
    json.dump(summary, open(prefix + "_variants_summary.json", "w"), indent=1)
    for k in summary["kernels"]:
        print(json.dumps(k))


if __name__ == "__main__":
    main(*sys.argv[1:3])
