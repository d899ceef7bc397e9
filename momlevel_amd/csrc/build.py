"""Build libmomlevel_hip.so (gfx950) in-tree with hipcc.

    python -m momlevel_amd.csrc.build [--force]

hipcc cross-compiles without a GPU.  The .so lands in momlevel_amd/ so that it
travels with the source tree (it is git-ignored, not gpurun-ignored).
``-ffp-contract=off`` is REQUIRED: the kernels reproduce the reference's numpy
arithmetic operator for operator (no FMA contraction); see csrc/eos_device.hpp.
"""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SOURCES = [os.path.join(HERE, "momlevel_hip.hip"), os.path.join(HERE, "momlevel_promote.hip"),
           os.path.join(HERE, "momlevel_strat.hip"), os.path.join(HERE, "host_copy.cpp")]
DEPENDS = SOURCES + [
    os.path.join(HERE, "eos_device.hpp"),
    os.path.join(HERE, "eos_promote.hpp"),
    os.path.join(HERE, "mlx_internal.hpp"),
    os.path.join(ROOT, "include", "momlevel_hip.h"),
    os.path.abspath(__file__),
]
LIB = os.path.join(PKG, "libmomlevel_hip.so")

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-ffp-contract=off",
    "-fPIC",
    "-shared",
    "-std=c++17",
    "-Wall",
    "-Wno-unused-variable",
    "-Wno-sometimes-uninitialized",
]


# What the timed kernels are built from: the sources of their translation units, the headers
# those include, the public header and the compiler flags.  bench.py and the profile summaries
# stamp this sha on their numbers; a committed counter profile is quoted only while it matches.
TIMED_SOURCES = [
    os.path.join(HERE, "momlevel_hip.hip"),
    os.path.join(HERE, "eos_device.hpp"),
    os.path.join(HERE, "mlx_internal.hpp"),
    os.path.join(HERE, "momlevel_promote.hip"),
    os.path.join(HERE, "eos_promote.hpp"),
    os.path.join(ROOT, "include", "momlevel_hip.h"),
]


def source_sha(paths=None, flags=True):
    """sha256 (first 16 hex digits) of ``paths`` (default TIMED_SOURCES) and, with ``flags``, of
    the compiler flags"""
    import hashlib

    h = hashlib.sha256()
    for path in (TIMED_SOURCES if paths is None else paths):
        with open(path, "rb") as f:
            h.update(f.read())
    if flags:
        h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def strat_source_sha():
    """the stratification kernels' own guard: csrc/momlevel_strat.hip (+ what it includes, + flags)"""
    return source_sha([os.path.join(HERE, "momlevel_strat.hip"), os.path.join(HERE, "eos_device.hpp"),
                       os.path.join(HERE, "mlx_internal.hpp")])


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")
    return exe


def is_stale():
    if not os.path.exists(LIB):
        return True
    lib_mtime = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > lib_mtime for p in DEPENDS)


def build(force=False, verbose=False, extra_flags=()):
    """Compile the library if it is missing or older than its sources; return its path."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc()] + FLAGS + list(extra_flags) + SOURCES + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr:
        print(res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
