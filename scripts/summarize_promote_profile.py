#!/usr/bin/env python3
"""Condense scripts/run_promote_profile.sh's rocprofv3 passes into profiles/<tag>_promote_summary.json
and profiles/<tag>_promote_kernel_stats.csv.

    python scripts/summarize_promote_profile.py gpurun_out/prof_r03p profiles/r03 [--nt 16]

Per kernel of the any-dtype paths (k_eos_promote<...>, the mixed-dtype generic K1 / K2 twins, and
the tuned kernels the probe times beside them): calls, mean duration (--kernel-trace --stats), HBM
bytes per launch from the counter passes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (KiB counters;
gfx950 reports half of a wide coalesced read -- MI355X_MICROARCH.md, HBM), the algorithmic bytes of
the probe's case and their ratio.  Warm-up and timed launches of a case are the same kernel on the
same operands; launches of a kernel on smaller grids (one slab) are left out of the means."""
import csv
import json
import os
import sys
from collections import defaultdict

KEEP = ("k_eos_promote", "k_eos_map", "k_steric_global", "k_steric_local")
# algorithmic bytes per cell of each probe case, by kernel name fragment (scripts/promote_probe.py)
ALGO = {
    "k_eos_promote<float, float, mlx::np::Weak, 4>": 12,  # density / alpha: 4+4 read, 4 written
    "k_eos_promote<float, float, float, 4>": 16,
    "k_eos_promote<float, double, double, 2>": 28,
    "k_eos_promote<double, double, double, 2>": 32,
    "k_eos_map<double": 24,
    "k_steric_global<double, 1, 4, 0, 3,": 12,
    "k_steric_global<double, 1, 4, 0, 4,": 12,
    "k_steric_global<double, 2, 4, 0, 0,": 16,
    "k_steric_global<float, 4, 2, 0, 1,": 8,
    "k_steric_local<double, 1, 8, 0, 3,": 20,
    "k_steric_local<double, 1, 8, 0, 4,": 20,
    "k_steric_local<double, 2, 16, 0, 0,": 24,
    "k_steric_local<float, 4, 6, 0, 1,": 16,
}


def short(name):
    name = name.replace("void mlx::", "")
    return name.split("(")[0]


def full_size(groups):
    """{kernel: [(grid, value)]} -> {kernel: values of the dispatches with that kernel's LARGEST
    grid}: the probe also launches some kernels on one (z,y,x) slab (the reference density of the
    K2 cases); only the full-record launches are the probe's cases."""
    out = {}
    for k, items in groups.items():
        top = max(g for g, _ in items)
        out[k] = [v for g, v in items if g == top]
    return out


def counter_means(path, counter):
    groups = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not any(k in r["Kernel_Name"] for k in KEEP):
            continue
        groups[short(r["Kernel_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return {k: sum(v) / len(v) for k, v in full_size(groups).items()}


def main(src, prefix, nt=16):
    cells = nt * 75 * 1080 * 1440
    fetch = counter_means(os.path.join(src, "pmc_fetch", "run_counter_collection.csv"), "FETCH_SIZE")
    write = counter_means(os.path.join(src, "pmc_write", "run_counter_collection.csv"), "WRITE_SIZE")
    groups, meta = defaultdict(list), {}
    for r in csv.DictReader(open(os.path.join(src, "trace", "run_kernel_trace.csv"))):
        if not any(k in r["Kernel_Name"] for k in KEEP):
            continue
        k = short(r["Kernel_Name"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        groups[k].append((grid, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"])}
    rows = []
    for k, ms_list in sorted(full_size(groups).items(), key=lambda kv: kv[0]):
        ms = sum(ms_list) / len(ms_list)
        traffic = (2 * fetch[k] + write[k]) * 1024 if k in fetch and k in write else None
        bpc = next((v for frag, v in ALGO.items() if frag in k), None)
        if bpc is None:  # launched on single slabs only (reference densities): not a probe case
            continue
        row = {"kernel": k, "full_record_launches": len(ms_list), "mean_ms": round(ms, 4),
               "min_ms": round(min(ms_list), 4), "max_ms": round(max(ms_list), 4), **meta[k],
               "Gcells_per_s": round(cells / ms / 1e6, 1)}
        if bpc is not None:
            row["algorithmic_bytes_per_cell"] = bpc
            row["algorithmic_GB_per_s"] = round(bpc * cells / ms / 1e6, 1)
            row["frac_of_8TBs"] = round(bpc * cells / ms / 1e6 / 8000.0, 4)
        if traffic is not None:
            row["hbm_GB_per_launch"] = round(traffic / 1e9, 3)
            row["hbm_bytes_per_cell"] = round(traffic / cells, 3)
            if bpc is not None:
                row["traffic_over_algorithmic"] = round(traffic / (bpc * cells), 4)
        rows.append(row)
    out = {"source": "scripts/run_promote_profile.sh -> scripts/promote_probe.py --nt %d" % nt,
           "cells_per_launch": cells, "note": __doc__.split("Per kernel")[1].strip().replace("\n", " "),
           "kernels": rows}
    json.dump(out, open(prefix + "_promote_summary.json", "w"), indent=1)
    with open(os.path.join(src, "trace", "run_kernel_stats.csv")) as f, \
            open(prefix + "_promote_kernel_stats.csv", "w", newline="") as g:
        reader = csv.DictReader(f)
        w = csv.DictWriter(g, fieldnames=reader.fieldnames)
        w.writeheader()
        w.writerows(r for r in reader if any(k in r["Name"] for k in KEEP))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    nt = int(sys.argv[sys.argv.index("--nt") + 1]) if "--nt" in sys.argv else 16
    main(sys.argv[1], sys.argv[2], nt)
