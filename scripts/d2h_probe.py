#!/usr/bin/env python3
"""Where does the result download of a local call spend its time?  Page-locked allocation of the
result arrays (torch's pinned allocator -> hipHostMalloc), the device -> page-locked copy itself,
and the alternative: device -> staging ring -> pageable numpy array (8 memcpy threads).

    python scripts/d2h_probe.py [--gb 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import hostio  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=8.0)
    a = ap.parse_args()
    n = int(a.gb * 1e9 / 8)
    dev = torch.rand(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    out = {"bytes_GB": round(n * 8 / 1e9, 2)}
    for rep in range(2):
        t0 = time.perf_counter()
        host = hostio.pinned_array((n,), np.float64)
        t1 = time.perf_counter()
        hostio.download_into(host, dev)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[f"pinned_alloc_s_{rep}"] = round(t1 - t0, 3)
        out[f"d2h_into_pinned_s_{rep}"] = round(t2 - t1, 3)
        out[f"d2h_into_pinned_GB/s_{rep}"] = round(n * 8 / (t2 - t1) / 1e9, 1)
        # a second copy into the SAME (now resident) pinned array
        t3 = time.perf_counter()
        hostio.download_into(host, dev)
        torch.cuda.synchronize()
        out[f"d2h_into_pinned_again_GB/s_{rep}"] = round(n * 8 / (time.perf_counter() - t3) / 1e9, 1)
        del host
    limit = hostio.PINNED_RESULT_LIMIT
    hostio.PINNED_RESULT_LIMIT = 0  # pageable result arrays, filled through the staging ring
    for rep in range(2):
        t0 = time.perf_counter()
        host = hostio.pinned_array((n,), np.float64)
        t1 = time.perf_counter()
        hostio.download_into(host, dev)
        t2 = time.perf_counter()
        out[f"pageable_alloc_s_{rep}"] = round(t1 - t0, 3)
        out[f"d2h_through_ring_s_{rep}"] = round(t2 - t1, 3)
        out[f"d2h_through_ring_GB/s_{rep}"] = round(n * 8 / (t2 - t1) / 1e9, 1)
        t3 = time.perf_counter()
        hostio.download_into(host, dev)
        out[f"d2h_through_ring_again_GB/s_{rep}"] = round(n * 8 / (time.perf_counter() - t3) / 1e9, 1)
        del host
    hostio.PINNED_RESULT_LIMIT = limit
    # H2D for comparison (through the ring, from a pageable array)
    src = np.ones(n, dtype=np.float64)
    t0 = time.perf_counter()
    d2 = hostio.to_device(src, "cuda")
    torch.cuda.synchronize()
    out["h2d_through_ring_GB/s"] = round(n * 8 / (time.perf_counter() - t0) / 1e9, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
