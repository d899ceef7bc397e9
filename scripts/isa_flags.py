#!/usr/bin/env python3
"""ISA-level guard for the EXEC-masking inline asm (eos_device.hpp add_skipna / accumulate).

    python scripts/isa_flags.py [file.s ...]        # default: compiles momlevel_hip.hip -S (device)

Those asm blocks write VCC (v_cmp_o_f64 vcc, ...) and SCC (s_and_saveexec_b64) behind the
compiler's back.  They say so in their clobber lists -- but a clobber list is a promise the source
makes; round 3 shipped without "scc" in it and round 4's kernels then had an `s_cmp_eq_u32 ...
s_cselect_b64` pair scheduled AROUND five such blocks (profiles/r04_isa_scc_clobber_bug.txt: wrong
sums, no crash).  tests/test_static_names.py greps the clobber strings; this looks at what the
compiler actually emitted:

  for every kernel, over its control-flow graph, a scalar flag (SCC, VCC) is DIRTY from the end of
  an inline-asm block that writes it until the next compiler-emitted instruction that writes it;
  an instruction that READS the flag while it is dirty consumes a value the compiler never
  computed -- a violation.

Forward may-analysis: a block's entry state is the union of its predecessors' exit states
(fallthrough and branch targets), iterated to the fixed point.  Writer / reader tables are the
gfx9 SALU / VOP semantics as far as these kernels use them; an SALU opcode that is not listed counts
as NOT writing SCC (errs towards reporting).
"""

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "momlevel_amd", "csrc", "momlevel_hip.hip")

# SALU opcodes (prefix match on the mnemonic without its type suffix) that write SCC
SCC_WRITERS = (
    "s_cmp_", "s_cmpk_", "s_bitcmp", "s_add_", "s_sub_", "s_addc_", "s_subb_", "s_addk_",
    "s_min_", "s_max_", "s_and_", "s_or_", "s_xor_", "s_andn2_", "s_orn2_", "s_nand_", "s_nor_",
    "s_xnor_", "s_lshl_", "s_lshr_", "s_ashr_", "s_bfe_", "s_absdiff_", "s_abs_", "s_not_",
    "s_wqm_", "s_quadmask_", "s_bcnt0_", "s_bcnt1_", "s_lshl1_add", "s_lshl2_add", "s_lshl3_add",
    "s_lshl4_add", "s_andn1_", "s_orn1_",
)
SCC_READERS = ("s_cselect_", "s_cbranch_scc", "s_addc_", "s_subb_", "s_cmov_", "s_cmovk_")
# VALU opcodes that read VCC without naming it
VCC_IMPLICIT_READERS = ("v_div_fmas_",)
TERMINATORS = ("s_endpgm", "s_branch", "s_setpc_b64", "s_trap")


def compile_asm(path, src=SRC, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
           "-std=c++17", "--cuda-device-only", "-S", "-o", path, src] + list(extra)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
        raise RuntimeError(res.stderr[-4000:])
    return path


def kernels(lines):
    """-> {mangled name: [lines of the body]}"""
    out, cur = {}, None
    for line in lines:
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            cur = []
            out[m.group(1)] = cur
            continue
        if cur is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                cur = None
                continue
            cur.append(line.rstrip("\n"))
    return out


def _operands(text):
    return [o.strip() for o in text.split(",")] if text.strip() else []


def _names_vcc(op):
    return op.split("[")[0] in ("vcc", "vcc_lo", "vcc_hi")


def effects(instr):
    """(reads_scc, writes_scc, reads_vcc, writes_vcc) of one compiler-emitted instruction"""
    parts = instr.split(None, 1)
    op = parts[0]
    ops = _operands(parts[1].split(";")[0]) if len(parts) > 1 else []
    reads_scc = op.startswith(SCC_READERS)
    writes_scc = op.startswith(SCC_WRITERS) or "saveexec" in op
    # destinations: the first operand; a *_co_* VALU op's carry-out is its second operand
    dests = ops[:1]
    if re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_", op) or op.startswith(("v_div_scale_",
                                                                                 "v_mad_u64_u32",
                                                                                 "v_mad_i64_i32")):
        dests = ops[:2]
    if op.startswith(("s_cbranch", "s_cmp", "s_bitcmp", "s_waitcnt", "s_nop", "v_cmpx")) \
            or op.startswith(("global_store", "flat_store", "buffer_store", "ds_write", "ds_store",
                              "scratch_store")):
        dests = []
    if op.startswith("v_cmp_") and ops and not _names_vcc(ops[0]) and op.endswith("_e32"):
        dests = []  # (the e32 compare writes VCC implicitly: handled below)
    writes_vcc = any(_names_vcc(d) for d in dests)
    if op.startswith("v_cmp_") and op.endswith("_e32"):
        writes_vcc = True
    sources = ops[len(dests):] if dests else ops
    reads_vcc = any(_names_vcc(s) for s in sources) or op.startswith(VCC_IMPLICIT_READERS) \
        or op.startswith(("s_cbranch_vccz", "s_cbranch_vccnz"))
    return reads_scc, writes_scc, reads_vcc, writes_vcc


def asm_block_effects(block):
    """what the instructions of one inline-asm block write"""
    w_scc = w_vcc = False
    for line in block:
        instr = line.strip()
        if not instr or instr.startswith(";"):
            continue
        _, ws, _, wv = effects(instr)
        w_scc |= ws
        w_vcc |= wv
    return w_scc, w_vcc


def check_kernel(body):
    """-> [(line index, instruction, flag)] of reads of a flag last written by an inline-asm block"""
    # ---- basic blocks: items are ("i", idx, text) or ("asm", idx, [lines])
    blocks, labels, cur = [], {}, []

    def close():
        nonlocal cur
        blocks.append(cur)
        cur = []

    i = 0
    while i < len(body):
        line = body[i]
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            if cur:
                close()
            labels[m.group(1)] = len(blocks)
            i += 1
            continue
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            j = i + 1
            while j < len(body) and not body[j].strip().startswith(";;#ASMEND"):
                j += 1
            cur.append(("asm", i, body[i + 1:j]))
            i = j + 1
            continue
        if not line.startswith("\t") or s.startswith((";", ".")) or not s:
            i += 1
            continue
        cur.append(("i", i, s))
        if s.split()[0].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc_b64")):
            close()
        i += 1
    if cur:
        close()
    # ---- successors
    succ = []
    for b, items in enumerate(blocks):
        out = []
        last = items[-1] if items else None
        fall = True
        if last and last[0] == "i":
            op = last[2].split()[0]
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", last[2])
            if m and m.group(1) in labels:
                out.append(labels[m.group(1)])
            if op.startswith(TERMINATORS):
                fall = False
        if fall and b + 1 < len(blocks):
            out.append(b + 1)
        succ.append(out)

    def run(b, state, report):
        scc, vcc = state
        for kind, idx, payload in blocks[b]:
            if kind == "asm":
                ws, wv = asm_block_effects(payload)
                scc, vcc = scc or ws, vcc or wv
                continue
            rs, ws, rv, wv = effects(payload)
            if rs and scc and report is not None:
                report.append((idx, payload, "SCC"))
            if rv and vcc and report is not None:
                report.append((idx, payload, "VCC"))
            if ws:
                scc = False
            if wv:
                vcc = False
        return scc, vcc

    entry = [(False, False)] * len(blocks)
    work = list(range(len(blocks)))
    while work:
        b = work.pop()
        out = run(b, entry[b], None)
        for s in succ[b]:
            merged = (entry[s][0] or out[0], entry[s][1] or out[1])
            if merged != entry[s]:
                entry[s] = merged
                work.append(s)
    found = []
    for b in range(len(blocks)):
        run(b, entry[b], found)
    return found


def check_file(path, only=None):
    """-> ({kernel: violations}, kernels checked, inline-asm blocks seen)"""
    with open(path) as f:
        ks = kernels(f.readlines())
    bad, nasm = {}, 0
    for name, body in ks.items():
        if only and not any(o in name for o in only):
            continue
        nasm += sum(1 for line in body if line.strip().startswith(";;#ASMSTART"))
        v = check_kernel(body)
        if v:
            bad[name] = [(idx, ins, flag) for idx, ins, flag in v]
    return bad, len(ks), nasm


def main():
    paths = sys.argv[1:]
    if not paths:
        import tempfile

        out = os.path.join(tempfile.gettempdir(), "momlevel_amd_isa", "momlevel_hip.s")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        paths = [compile_asm(out)]
    rc = 0
    for p in paths:
        bad, n, nasm = check_file(p)
        print(f"{p}: {n} kernels, {nasm} inline-asm blocks, {len(bad)} kernels with violations")
        for k, v in bad.items():
            rc = 1
            print(f"  {k}: {len(v)}")
            for idx, ins, flag in v[:5]:
                print(f"      line {idx}: {ins}   <- reads {flag} last written by an inline-asm block")
    return rc


if __name__ == "__main__":
    sys.exit(main())
