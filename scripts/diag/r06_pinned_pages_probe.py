#!/usr/bin/env python3
"""Why do some page-locked staging buffers take device->host DMA at 26-29 GB/s and others at 56?
(scripts/diag/r06_numa_probe.py: not the NUMA node.)  Hypothesis: the page size behind the buffer --
2 MiB transparent huge pages give the IOMMU 2 MiB mappings, 4 KiB pages 16384 mappings per 64 MiB and
an IOTLB miss every page of a posted write stream.  Buffers by three routes, each a set of four used
round robin for 2 GiB of D2H and of H2D: torch's pinned allocator (hipHostMalloc); our own anonymous
mapping advised MADV_HUGEPAGE, faulted in, then page-locked in place (hipHostRegister through torch's
cudart binding); the same advised MADV_NOHUGEPAGE.  AnonHugePages of each mapping from smaps."""
import ctypes
import json
import mmap
import time

import numpy as np
import torch

PIECE = 64 << 20
HUGE = 2 << 20
dev = torch.device("cuda", 0)
src = torch.empty(PIECE, dtype=torch.uint8, device=dev)
dst = torch.empty(PIECE, dtype=torch.uint8, device=dev)
rt = torch.cuda.cudart()
torch.cuda.synchronize()


def huge_kib(addr):
    """AnonHugePages (KiB) of the mapping that holds addr"""
    cur = None
    with open("/proc/self/smaps") as f:
        for line in f:
            head = line.split()
            if "-" in head[0] and len(head) >= 5 and all(c in "0123456789abcdef-" for c in head[0]):
                lo, hi = (int(x, 16) for x in head[0].split("-"))
                cur = lo <= addr < hi
            elif cur and line.startswith("AnonHugePages:"):
                return int(head[1])
    return None


def own(kind):
    m = mmap.mmap(-1, PIECE + HUGE, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    a = np.frombuffer(m, dtype=np.uint8)
    off = (-a.ctypes.data) % HUGE
    m.madvise(mmap.MADV_HUGEPAGE if kind == "huge" else mmap.MADV_NOHUGEPAGE, off, PIECE)
    a = a[off:off + PIECE]
    a[:] = 1  # fault in
    t = torch.from_numpy(a)
    rc = rt.cudaHostRegister(t.data_ptr(), PIECE, 0)
    assert int(rc) == 0, rc
    return t, m


def rate(bufs, fn):
    for b in bufs:
        fn(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(32):
        fn(bufs[i % 4])
    torch.cuda.synchronize()
    return round(32 * PIECE / (time.perf_counter() - t0) / 1e9, 1)


rows = []
keep = []
for rep in range(8):
    for kind in ("torch_pinned", "own_huge", "own_small"):
        if kind == "torch_pinned":
            size = PIECE + 4096 * (1 + len(rows))  # (a size of its own: no block from torch's cache)
            bufs = [torch.empty(size, dtype=torch.uint8, pin_memory=True)[:PIECE] for _ in range(4)]
            for b in bufs:
                b.fill_(1)
        else:
            made = [own("huge" if kind == "own_huge" else "small") for _ in range(4)]
            bufs = [t for t, _ in made]
            keep.append(made)
        rows.append({"kind": kind,
                     "AnonHugePages_KiB": [huge_kib(b.data_ptr()) for b in bufs],
                     "d2h_GB/s": rate(bufs, lambda b: b.copy_(src, non_blocking=True)),
                     "h2d_GB/s": rate(bufs, lambda b: dst.copy_(b, non_blocking=True))})
        keep.append(bufs)
out = {}
for kind in ("torch_pinned", "own_huge", "own_small"):
    out[kind] = {"d2h_GB/s": [r["d2h_GB/s"] for r in rows if r["kind"] == kind],
                 "h2d_GB/s": [r["h2d_GB/s"] for r in rows if r["kind"] == kind],
                 "AnonHugePages_KiB_first_buffer": [r["AnonHugePages_KiB"][0] for r in rows if r["kind"] == kind]}
print(json.dumps(out))
