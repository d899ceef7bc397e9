#!/usr/bin/env python3
"""Static VALU-instruction count of the kernels' hot loops from the gfx950 assembly.

    python scripts/isa_count.py [--filter k_steric_global] [--dump NAME_SUBSTRING]

Compiles momlevel_hip.hip to assembly (device side only), finds every loop of every kernel (a
backward branch to a label), and prints for the LARGEST loop of each kernel the number of VALU
instructions (v_*), of which f64 / packed-f32 / transcendental, the memory instructions, and the
VGPR / occupancy lines of the kernel descriptor.  A CPU-side proxy for rocprofv3's SQ_INSTS_VALU:
cells per loop iteration are known per kernel (K1: 8 per thread and time step), so
VALU per cell = count / cells.  Used to evaluate instruction-selection changes (the guarded
division, the skipna accumulate) before spending GPU time.
"""

import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "momlevel_amd", "csrc", "momlevel_hip.hip")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return out.stdout.splitlines()
    except OSError:
        return names


def compile_asm(path, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
           "-std=c++17", "--cuda-device-only", "-S", "-o", path, SRC] + list(extra)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
        sys.exit(res.stderr)


def kernels(asm):
    """-> {mangled name: [lines]} for every kernel body"""
    out, cur, name = {}, None, None
    for line in asm:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                cur = None
                continue
            cur.append(line.rstrip("\n"))
    return out


def loops(body):
    """[(start index, end index)] of every backward branch"""
    labels = {}
    for i, line in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            labels[m.group(1)] = i
    found = []
    for i, line in enumerate(body):
        m = re.match(r"^\ts_cbranch_\w+ (\.LBB\w+)", line) or re.match(r"^\ts_branch (\.LBB\w+)", line)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            found.append((labels[m.group(1)], i))
    return found


def classify(lines):
    """instruction mix of a loop body; the never-taken IEEE-division fallback blocks (the basic
    blocks that carry the kernels' asm marker, eos_device.hpp quotients<>) are left out and
    counted as `cold`"""
    c = dict(valu=0, f64=0, pk=0, trans=0, cndmask=0, cmp=0, mov=0, vmem=0, salu=0, lds=0, cvt=0,
             cold=0)
    blocks, cur = [], []
    for line in lines:
        if re.match(r"^\.LBB\w+:", line) or re.match(r"^; %bb\.\d+:", line):
            blocks.append(cur)
            cur = []
        cur.append(line)
    blocks.append(cur)
    for blk in blocks:
        is_cold = any("; IEEE-division fallback" in x for x in blk)
        for line in blk:
            m = re.match(r"^\t([a-z_0-9]+)", line)
            if not m:
                continue
            if is_cold:
                c["cold"] += 1
                continue
            op = m.group(1)
            if op.startswith("v_"):
                c["valu"] += 1
                if "f64" in op:
                    c["f64"] += 1
                if op.startswith("v_pk_"):
                    c["pk"] += 1
                if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp")):
                    c["trans"] += 1
                if op.startswith("v_cndmask"):
                    c["cndmask"] += 1
                if op.startswith("v_cmp"):
                    c["cmp"] += 1
                if op.startswith(("v_mov", "v_accvgpr")):
                    c["mov"] += 1
                if op.startswith("v_cvt"):
                    c["cvt"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
    return c


def meta(body_and_tail):
    out = {}
    for line in body_and_tail:
        for key in ("NumVgprs", "NumAgprs", "Occupancy", "ScratchSize", "TotalNumVgprs"):
            m = re.match(rf"^; {key}: (\d+)", line)
            if m:
                out[key] = int(m.group(1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="k_steric")
    ap.add_argument("--dump", default=None, help="print the hot loop of kernels whose demangled name contains this")
    ap.add_argument("--asm", default="/tmp/isa/mlx.s")
    ap.add_argument("--no-compile", action="store_true")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.asm), exist_ok=True)
    if not a.no_compile:
        compile_asm(a.asm, ["-D" + d for d in a.D])
    with open(a.asm) as f:
        asm = f.readlines()
    ks = kernels(asm)
    names = list(ks)
    pretty = dict(zip(names, demangle(names)))
    # the metadata comment block follows each kernel: find it in the full text
    text = "".join(asm)
    for name in names:
        p = pretty[name]
        if a.filter not in p:
            continue
        body = ks[name]
        ls = loops(body)
        if not ls:
            continue
        # hot loop = the one with most VALU instructions
        best = max(ls, key=lambda se: classify(body[se[0]:se[1] + 1])["valu"])
        c = classify(body[best[0]:best[1] + 1])
        i = text.find(f"\n{name}:")
        j = text.find("; Occupancy", i)
        m = meta(text[i:j + 40].splitlines())
        short = re.sub(r"^void mlx::", "", p).split("(")[0]
        print(f"{short:72s} loop VALU {c['valu']:4d} (f64 {c['f64']:3d} pk {c['pk']:3d} trans {c['trans']:2d} "
              f"cnd {c['cndmask']:3d} cmp {c['cmp']:3d} mov {c['mov']:3d} cvt {c['cvt']:3d}) "
              f"vmem {c['vmem']:2d} salu {c['salu']:3d} lds {c['lds']:2d} cold {c['cold']:3d} | vgpr {m.get('NumVgprs')} "
              f"agpr {m.get('NumAgprs')} occ {m.get('Occupancy')} scratch {m.get('ScratchSize')}")
        if a.dump and a.dump in p:
            print("\n".join(body[best[0]:best[1] + 1]))


if __name__ == "__main__":
    main()
