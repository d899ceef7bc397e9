#!/usr/bin/env python3
"""Markdown rows for DESIGN.md section 5 straight from a bench.py JSON line (no retyping of numbers).

    python scripts/bench_table.py gpurun_out/r04_bench2.log
"""
import json
import sys


def row(label, r, extra=""):
    cols = [label, f"{r['ms']:.2f} ms", f"{r['Mcells/s'] / 1e3:.0f} Gcells/s", f"{r['frac_of_8TBs']:.3f}"]
    cols.append(f"{r['frac_of_matching_probe']:.3f}" if "frac_of_matching_probe" in r else
                f"{r['frac_of_read_write_probe']:.3f}" if "frac_of_read_write_probe" in r else "")
    cols.append(f"{r['valu_instr_per_cell']:.1f}" if "valu_instr_per_cell" in r else "")
    cols.append(f"{r['frac_of_f64_fma_probe']:.2f}" if "frac_of_f64_fma_probe" in r else "")
    cols.append(r.get("kernel", "") + extra)
    return "| " + " | ".join(cols) + " |"


def main(path):
    line = [l for l in open(path) if l.startswith("{")][-1]
    d = json.loads(line)
    rf = d["roofline"]
    print(f"value {d['value'] / 1e3:.1f} Gcells/s, {d['ms_per_step']} ms/step; K1 launch {rf['launch_ms']} ms, "
          f"{rf['achieved']} GB/s = {rf['frac']} of 8 TB/s; traffic {rf['traffic']} GB/launch; kernel {rf['kernel']}; "
          f"valu/cell {rf.get('valu_instr_per_cell')} frac_of_f64_fma_probe {rf.get('frac_of_f64_fma_probe')}")
    print("valu_roofline:", {k: v for k, v in d["valu_roofline"].items() if k != "definition"})
    print("stream_read_probe", d.get("stream_read_probe"), "read_write_probe", d.get("stream_read_write_probe"))
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline_processes"].get("value"),
          d["cpu_baseline_fused_openmp"]["value"], d["cpu_baseline"]["cpu"], "parity", d["parity"])
    print("land_skipping", d.get("land_skipping"))
    print("| case | time | rate | of 8 TB/s | of matching probe | VALU instr/cell | of f64 fma probe | kernel |")
    print("|---|---|---|---|---|---|---|---|")
    for k, v in d.items():
        if isinstance(v, dict) and "ms" in v and "Mcells/s" in v:
            print(row(k, v, f" slab={v['slab_bit_identical_to_oracle']}" if "slab_bit_identical_to_oracle" in v else ""))
    f = d.get("config5_f32")
    if f:
        print("float32:")
        for mode in ("default", "faithful", "faithful_fused", "upcast", "upcast_fused"):
            for k, v in f[mode].items():
                if isinstance(v, dict) and "ms" in v and "Mcells/s" in v:
                    if mode != "default" and k in f["default"] and f["default"][k] is v:
                        continue
                    print(row(f"{mode}.{k}", v, f" slab={v['slab_bit_identical_to_oracle']}"
                              if "slab_bit_identical_to_oracle" in v else ""))
        print("parity", f["parity"])


if __name__ == "__main__":
    main(sys.argv[1])
