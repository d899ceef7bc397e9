"""How fast can this host fill a page-locked staging buffer?  (hostio.py's host-side copy.)

    python scripts/host_copy_probe.py

Prints the CPU resources the process sees (visible cores, affinity, cgroup quota, torch's thread
count) and the rate of: torch's CPU copy_, numpy copyto, libc memcpy from 1..16 Python threads
(ctypes releases the GIL) -- pageable source, page-locked destination, 256 MiB -- and the H2D rate
from the page-locked buffer.  Decides hostio._host_copy's strategy; profiles/r03_host_copy_probe.log.
"""
import ctypes
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import hostio  # noqa: E402


def rate(fn, n, reps=5):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return n / ((time.perf_counter() - t) / reps) / 1e9


def main():
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)),
          "torch threads", torch.get_num_threads(), "hostio.host_threads", hostio.host_threads())
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            print(f, open(f).read().strip())
        except OSError:
            pass
    n = 256 << 20
    src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
    pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    plain = torch.empty(n, dtype=torch.uint8)
    ts = torch.from_numpy(src)
    for name, dst in (("pinned", pinned), ("pageable", plain)):
        print(f"-- destination: {name}")
        d64, s64 = dst.view(torch.float64), ts.view(torch.float64)
        print(f"torch copy_ float64 view ({torch.get_num_threads()} threads): {rate(lambda: d64.copy_(s64), n):6.1f} GB/s")
        dn = dst.numpy()
        print(f"numpy copyto:                              {rate(lambda: np.copyto(dn, src), n):6.1f} GB/s")
        memcpy = ctypes.CDLL(None).memcpy
        memcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        memcpy.restype = ctypes.c_void_p
        d, s = dst.data_ptr(), ts.data_ptr()
        for th in (1, 2, 4, 8, 16, 32):
            pool = ThreadPoolExecutor(th)
            step = n // th

            def par():
                fs = [pool.submit(memcpy, d + o, s + o, step) for o in range(0, n, step)]
                for f in fs:
                    f.result()

            print(f"libc memcpy, {th:2d} threads:                   {rate(par, n):6.1f} GB/s")
            pool.shutdown()
        for th in (8, 16):
            old = torch.get_num_threads()
            torch.set_num_threads(th)
            print(f"torch copy_ with set_num_threads({th:2d}):       {rate(lambda: d64.copy_(s64), n):6.1f} GB/s")
            torch.set_num_threads(old)
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")

    def h2d():
        dev.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()

    print(f"H2D from the page-locked buffer:            {rate(h2d, n):6.1f} GB/s")
    a = np.random.default_rng(1).standard_normal(n // 8)
    big = torch.empty(n // 8, dtype=torch.float64, device="cuda")

    def staged():
        hostio.upload(torch.from_numpy(a), big)
        torch.cuda.synchronize()

    print(f"hostio.upload (staging ring, end to end):   {rate(staged, n):6.1f} GB/s")


if __name__ == "__main__":
    main()
