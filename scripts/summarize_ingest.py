"""Condense the rocprofv3 traces of scripts/ingest_profile.py into overlap figures.

    python scripts/summarize_ingest.py gpurun_out/prof_r03_ingest profiles/r03_ingest_overlap.json

For each roctx range "steric(domain=...)": wall time, the union of the H2D copies' busy time, of the
D2H copies', of the kernels', and their pairwise overlaps -- is the host link busy for (nearly) the
whole call (then the PCIe rate IS the ceiling of this row), and do kernels / downloads hide behind it?
"""
import csv
import glob
import json
import os
import sys


def one(src, pattern):
    hits = sorted(glob.glob(os.path.join(src, "**", pattern), recursive=True))
    return hits[0] if hits else None


def union(intervals):
    out, total = [], 0
    for a, b in sorted(intervals):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out, sum(b - a for a, b in out)


def overlap(u1, u2):
    i = j = tot = 0
    while i < len(u1) and j < len(u2):
        a, b = max(u1[i][0], u2[j][0]), min(u1[i][1], u2[j][1])
        if b > a:
            tot += b - a
        if u1[i][1] < u2[j][1]:
            i += 1
        else:
            j += 1
    return tot


def clip(intervals, lo, hi):
    return [(max(a, lo), min(b, hi)) for a, b in intervals if b > lo and a < hi]


def main(src, out_path):
    markers = list(csv.DictReader(open(one(src, "*marker_api_trace.csv"))))
    copies = list(csv.DictReader(open(one(src, "*memory_copy_trace.csv"))))
    kernels = list(csv.DictReader(open(one(src, "*kernel_trace.csv"))))
    ts = lambda r: (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))  # noqa: E731
    h2d = [ts(r) for r in copies if "HOST_TO_DEVICE" in r.get("Direction", "").upper()]
    d2h = [ts(r) for r in copies if "DEVICE_TO_HOST" in r.get("Direction", "").upper()]
    # the runtime moves device -> page-locked host with a blit KERNEL (__amd_rocclr_copyBuffer), not
    # with the DMA engines: those dispatches are downloads, not compute
    blit = [ts(r) for r in kernels if "rocclr_copy" in r["Kernel_Name"]]
    ker = [ts(r) for r in kernels if "rocclr_copy" not in r["Kernel_Name"]]
    d2h = d2h + blit
    out = {"source": src, "ranges": []}
    for r in markers:
        name = r.get("Function", "") or r.get("Name", "")
        if not name.startswith(("steric(domain=", "thermosteric(ds)")):
            continue
        lo, hi = ts(r)
        uh, th = union(clip(h2d, lo, hi))
        ud, td = union(clip(d2h, lo, hi))
        uk, tk = union(clip(ker, lo, hi))
        stage = [ts(m) for m in markers
                 if (m.get("Function", "") or m.get("Name", "")).startswith("stage+H2D")
                 and int(m["Start_Timestamp"]) >= lo and int(m["End_Timestamp"]) <= hi]
        wall = hi - lo
        out["ranges"].append({
            "range": name, "wall_ms": wall / 1e6,
            "h2d_busy_ms": th / 1e6, "h2d_busy_frac": th / wall, "h2d_copies": len(clip(h2d, lo, hi)),
            "d2h_busy_ms": td / 1e6, "d2h_busy_frac": td / wall,
            "d2h_note": "SDMA copies + __amd_rocclr_copyBuffer blit kernels (device -> page-locked host)",
            "kernel_busy_ms": tk / 1e6, "kernel_busy_frac": tk / wall,
            "h2d_or_d2h_busy_frac": union([tuple(x) for x in uh] + [tuple(x) for x in ud])[1] / wall,
            "kernel_hidden_behind_h2d_frac": (overlap(uk, uh) / tk) if tk else None,
            "d2h_hidden_behind_h2d_frac": (overlap(ud, uh) / td) if td else None,
            "host_staging_ranges": len(stage),
            "host_staging_ms_total": sum(b - a for a, b in stage) / 1e6,
        })
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
