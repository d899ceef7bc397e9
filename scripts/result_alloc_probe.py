#!/usr/bin/env python3
"""What does the FIRST TOUCH of a fresh result array cost on this host, by how the array was
allocated?  The staged result download copies into an ordinary numpy array; when the previous
call's results are still alive that array is new memory from the OS, and the copy threads take its
page faults.  Strategies: numpy's default (np.empty: madvise(MADV_HUGEPAGE) on large blocks),
an anonymous mmap with MADV_NOHUGEPAGE, one with MADV_HUGEPAGE, and -- for reference -- the same
array touched a second time.

    python scripts/result_alloc_probe.py [--gb 6] [--keep 2]
"""
import argparse
import ctypes
import json
import mmap
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import _lib  # noqa: E402

MADV_HUGEPAGE, MADV_NOHUGEPAGE = 14, 15


def thp_settings():
    out = {}
    for k in ("enabled", "defrag", "shmem_enabled"):
        try:
            out[k] = open(f"/sys/kernel/mm/transparent_hugepage/{k}").read().strip()
        except OSError as e:
            out[k] = str(e)
    return out


def alloc(kind, n):
    if kind == "numpy":
        return np.empty(n, dtype=np.uint8)
    m = mmap.mmap(-1, n, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    a = np.frombuffer(m, dtype=np.uint8)
    libc = ctypes.CDLL(None, use_errno=True)
    libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rc = libc.madvise(a.ctypes.data, n, MADV_HUGEPAGE if kind == "mmap_hugepage" else MADV_NOHUGEPAGE)
    assert rc == 0, ctypes.get_errno()
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=6.0)
    ap.add_argument("--keep", type=int, default=2, help="earlier results kept alive")
    a = ap.parse_args()
    n = int(a.gb * (1 << 30))
    copy = _lib.load().mlx_host_copy
    piece = 64 << 20
    src = np.ones(piece, dtype=np.uint8)  # (a staging piece: the source is re-read, as in the ring)

    def fill(dst):
        t0 = time.perf_counter()
        for off in range(0, n, piece):
            m = min(piece, n - off)
            assert copy(dst.ctypes.data + off, src.ctypes.data, m, 8, 1) == 0
        return time.perf_counter() - t0

    rep = {"GB": a.gb, "thp": thp_settings(), "kept_alive": a.keep}
    for kind in ("numpy", "mmap_nohugepage", "mmap_hugepage", "numpy"):
        alive, first, again = [], [], []
        for _ in range(a.keep + 2):
            t0 = time.perf_counter()
            dst = alloc(kind, n)
            t_alloc = time.perf_counter() - t0
            first.append(round(n / (t_alloc + fill(dst)) / 1e9, 1))
            again.append(round(n / fill(dst) / 1e9, 1))
            alive.append(dst)
            if len(alive) > a.keep:
                alive.pop(0)
        key = kind if kind not in rep else kind + "_again"
        rep[key + "_first_touch_GB/s"] = first
        rep[key + "_second_touch_GB/s"] = again
        del alive, dst
    print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
