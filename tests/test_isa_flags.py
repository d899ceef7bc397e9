"""CPU: what the compiler EMITTED around the EXEC-masking inline asm.

eos_device.hpp's add_skipna / accumulate write VCC and SCC inside asm blocks.  Rounds 3's binaries
shipped without "scc" in the clobber lists; round 4 then saw an `s_cmp_eq_u32 ... s_cselect_b64`
pair scheduled around five such blocks (wrong sums, no crash; profiles/r04_isa_scc_clobber_bug.txt).
tests/test_static_names.py greps the clobber strings; scripts/isa_flags.py checks the gfx950
assembly itself: no compiler-emitted instruction may read SCC / VCC while the last writer of that
flag on some path is an inline-asm block.
"""

import importlib.util
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_flags", os.path.join(ROOT, "scripts", "isa_flags.py"))
isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa)

ASM_BLOCK = """\t;;#ASMSTART
\tv_cmp_o_f64 vcc, v[10:11], v[42:43]
\ts_and_saveexec_b64 s[52:53], vcc
\tv_fma_f64 v[22:23], v[10:11], v[42:43], v[22:23]
\ts_mov_b64 exec, s[52:53]
\t;;#ASMEND"""


def _body(text):
    return text.split("\n")


def test_the_recorded_round4_hazard_is_flagged():
    with open(os.path.join(ROOT, "profiles", "r04_isa_scc_clobber_bug.txt")) as f:
        body = [line.rstrip("\n") for line in f if not line.startswith("#")]
    found = isa.check_kernel(body)
    assert [(ins, flag) for _, ins, flag in found] == [("s_cselect_b64 s[52:53], -1, 0", "SCC")]


def test_flag_tracking_on_small_cases():
    # a compare AFTER the block is the compiler's own: fine
    ok = _body(f"\ts_cmp_eq_u32 s1, 3\n\ts_cselect_b64 s[2:3], -1, 0\n{ASM_BLOCK}\n"
               "\ts_cmp_eq_u32 s1, 3\n\ts_cselect_b64 s[2:3], -1, 0\n\ts_endpgm")
    assert isa.check_kernel(ok) == []
    # the compare BEFORE, the consumer AFTER: SCC is the asm block's
    bad = _body(f"\ts_cmp_eq_u32 s1, 3\n{ASM_BLOCK}\n\ts_cbranch_scc1 .LBB0_2\n\ts_nop 0\n"
                ".LBB0_2:\n\ts_endpgm")
    assert [f for _, _, f in isa.check_kernel(bad)] == ["SCC"]
    # s_mov / v_* do not rewrite SCC: still dirty two instructions later
    bad = _body(f"{ASM_BLOCK}\n\ts_mov_b32 s4, 0\n\tv_add_f64 v[0:1], v[0:1], v[2:3]\n"
                "\ts_addc_u32 s5, s5, 0\n\ts_endpgm")
    assert [f for _, _, f in isa.check_kernel(bad)] == ["SCC"]
    # VCC: v_cndmask reading the block's vcc; a compiler v_cmp in between clears it
    bad = _body(f"{ASM_BLOCK}\n\tv_cndmask_b32_e32 v1, v2, v3, vcc\n\ts_endpgm")
    assert [f for _, _, f in isa.check_kernel(bad)] == ["VCC"]
    ok = _body(f"{ASM_BLOCK}\n\tv_cmp_gt_f64_e32 vcc, v[0:1], v[2:3]\n"
               "\tv_cndmask_b32_e32 v1, v2, v3, vcc\n\ts_endpgm")
    assert isa.check_kernel(ok) == []
    ok = _body(f"{ASM_BLOCK}\n\tv_add_co_u32_e32 v1, vcc, v2, v3\n"
               "\tv_addc_co_u32_e32 v4, vcc, v5, v6, vcc\n\ts_endpgm")
    assert isa.check_kernel(ok) == []
    # across the control-flow graph: dirty on ONE incoming path is enough (loop back edge)
    loop = _body("\ts_cmp_lg_u32 s0, 0\n.LBB0_1:\n\ts_cselect_b32 s1, 1, 0\n"
                 f"{ASM_BLOCK}\n\ts_sub_i32 s0, s0, 1\n\ts_cmp_lg_u32 s0, 0\n"
                 "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm")
    assert isa.check_kernel(loop) == []  # (s_sub / s_cmp rewrite SCC before the back edge)
    loop = _body("\ts_cmp_lg_u32 s0, 0\n.LBB0_1:\n\ts_cselect_b32 s1, 1, 0\n"
                 f"\ts_cmp_lg_u32 s0, 5\n{ASM_BLOCK}\n\ts_cbranch_vccz .LBB0_1\n\ts_endpgm")
    flags = sorted(f for _, _, f in isa.check_kernel(loop))
    assert flags == ["SCC", "VCC"]  # the back edge carries the block's SCC into s_cselect
    # code after an unconditional branch is not reached by fallthrough
    ok = _body(f"{ASM_BLOCK}\n\ts_branch .LBB0_9\n.LBB0_3:\n\ts_cselect_b32 s1, 1, 0\n"
               ".LBB0_9:\n\ts_endpgm")
    assert isa.check_kernel(ok) == []


def _asm_of(src_dir, out):
    """gfx950 assembly of momlevel_hip.hip as found under ``src_dir``, cached by source hash"""
    import hashlib

    h = hashlib.sha256()
    for name in sorted(os.listdir(os.path.join(src_dir, "momlevel_amd", "csrc"))):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(src_dir, "momlevel_amd", "csrc", name), "rb") as f:
                h.update(f.read())
    # (40 MB apiece: cached in the system's temporary directory, not in the tree -- a tree-local
    #  cache travelled to the GPU box with every gpurun snapshot)
    import tempfile

    path = os.path.join(tempfile.gettempdir(), "momlevel_amd_isa", f"{out}_{h.hexdigest()[:16]}.s")
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        isa.compile_asm(path + ".tmp", os.path.join(src_dir, "momlevel_amd", "csrc", "momlevel_hip.hip"))
        os.replace(path + ".tmp", path)
    return path


@pytest.mark.timeout(900)
def test_no_kernel_consumes_a_flag_written_by_the_exec_masking_asm(tmp_path):
    """HEAD: every kernel of momlevel_hip.hip, all ~17000 inline-asm blocks: clean.  The same
    sources with "scc" dropped from the clobber lists (round 3's state): the checker reports the
    kernels whose compare results the compiler then carried across a block -- the guard fails on
    the hazardous build, not only on its source text."""
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    bad_src = tmp_path / "noscc"
    os.makedirs(bad_src / "momlevel_amd" / "csrc")
    os.makedirs(bad_src / "include")
    csrc = os.path.join(ROOT, "momlevel_amd", "csrc")
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".hpp")):
            text = open(os.path.join(csrc, name)).read()
            if name == "eos_device.hpp":
                assert text.count(': "vcc", "scc")') >= 2
                text = text.replace(': "vcc", "scc")', ': "vcc")')
            (bad_src / "momlevel_amd" / "csrc" / name).write_text(text)
    shutil.copy(os.path.join(ROOT, "include", "momlevel_hip.h"), bad_src / "include")
    with ThreadPoolExecutor(2) as pool:  # the two compilations side by side (~75 s each)
        head = pool.submit(_asm_of, ROOT, "head")
        bad = pool.submit(_asm_of, str(bad_src), "noscc")
        head, bad = head.result(), bad.result()
    violations, nkernels, nasm = isa.check_file(head)
    assert nkernels > 100 and nasm > 10000
    assert violations == {}, {k: v[:2] for k, v in list(violations.items())[:3]}
    violations, _, _ = isa.check_file(bad)
    names = subprocess.run(["c++filt"], input="\n".join(violations), capture_output=True,
                           text=True).stdout
    assert len(violations) >= 4 and "k_steric_global" in names
    assert all(flag == "SCC" for v in violations.values() for _, _, flag in v)
