#!/usr/bin/env python3
"""Tuning builds of the library: one .so per set of -D overrides, under scripts/variants/ (built
artefacts: git-ignored, they travel to the GPU box with the snapshot).

    python scripts/build_variants.py nti64_12:-DMLX_TUNE_NTI64=12 nti32_4:-DMLX_TUNE_NTI32=4 ...

Each is then timed in its own process on one box: MOMLEVEL_AMD_LIB=scripts/variants/lib_<name>.so
python scripts/ab_k2.py (or ab_k1.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from momlevel_amd.csrc import build  # noqa: E402


def main():
    out = os.path.join(ROOT, "scripts", "variants")
    os.makedirs(out, exist_ok=True)
    jobs = []
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition(":")
        lib = os.path.join(out, f"lib_{name}.so")
        cmd = [build.hipcc()] + build.FLAGS + [f for f in flags.split(",") if f] + build.SOURCES + ["-o", lib]
        jobs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(jobs) >= 4:  # 8 CPUs, ~2 GB per hipcc
            name0, p = jobs.pop(0)
            text = p.communicate()[0]
            print(name0, "rc", p.returncode, text[-400:] if p.returncode else "")
    for name0, p in jobs:
        text = p.communicate()[0]
        print(name0, "rc", p.returncode, text[-400:] if p.returncode else "")


if __name__ == "__main__":
    main()
