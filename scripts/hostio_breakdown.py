#!/usr/bin/env python3
"""Where does the reference's recorded call (scripts/example_call.py) spend its wall time on the
host side?  Wraps hostio's staged upload / download and the host memcpy and reports, per thread,
the seconds inside each and the memcpy rate while the other direction is (or is not) running.

    python scripts/hostio_breakdown.py [--nt 60] [--nz 35]
"""
import argparse
import collections
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from momlevel_amd import hostio  # noqa: E402
import example_call  # noqa: E402

acc = collections.defaultdict(lambda: [0.0, 0, 0])  # name -> [seconds, bytes, calls]
lock = threading.Lock()


def timed(name, fn, nbytes_of):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            name_ = threading.current_thread().name
            who = ("upload-worker" if name_.startswith("mlx-upload") else
                   "download-worker" if name_.startswith("mlx-download") else "main")
            with lock:
                e = acc[f"{name}[{who}]"]
                e[0] += dt
                e[1] += nbytes_of(*a, **k)
                e[2] += 1
    return wrapper


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=60)
    ap.add_argument("--nz", type=int, default=35)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--case", default="example", choices=["example", "ingest32", "ingest64"],
                    help="example: the recorded thermosteric(ds) call; ingest*: scripts/ingest_check.py's "
                         "steric(d, domain='local') on 48 x 75 x 576 x 360 host fields")
    a = ap.parse_args()
    hostio._host_copy = timed("memcpy", hostio._host_copy, lambda d, s: d.numel())
    hostio.upload = timed("upload", hostio.upload,
                          lambda h, d, **k: h.numel() * h.element_size())
    hostio.download_into = timed("download_into", hostio.download_into,
                                 lambda o, d, *r: o.nbytes)
    hostio._enqueue_download = timed("enqueue_download", hostio._enqueue_download,
                                     lambda o, d, *r: o.nbytes)
    hostio.Downloader.submit = timed("Downloader.submit (blocked)", hostio.Downloader.submit,
                                     lambda self, pairs: sum(o.nbytes for o, _ in pairs))
    hostio.Downloader.finish = timed("Downloader.finish (blocked)", hostio.Downloader.finish,
                                     lambda self: 0)
    def reset():
        with lock:
            acc.clear()

    if a.case == "example":
        out2 = example_call.run(a.nt, a.nz, reps=a.reps, before_call=reset)  # counters: the last call only
    else:
        import numpy as np
        import torch

        import momlevel_amd as m
        from ingest_check import dataset

        d = dataset(48, 75, 576, 360, np.float32 if a.case == "ingest32" else np.float64)
        walls = []
        for _ in range(a.reps):
            reset()
            t0 = time.perf_counter()
            res, ref = m.steric(d, domain="local")
            torch.cuda.synchronize()
            walls.append(round(time.perf_counter() - t0, 3))
            del res, ref
        out2 = {"wall_s": walls}
    with lock:
        second = {k: list(v) for k, v in acc.items()}
    rep = {"case": a.case, "threads": hostio.host_threads(), "piece_MiB": hostio.PIECE_BYTES >> 20,
           "download_ring": os.environ.get("MOMLEVEL_AMD_DOWNLOAD_RING", "default"),
           "wall_s": out2["wall_s"]}
    for k, (s, b, c) in sorted(second.items()):
        rep[k] = {"s": round(s, 3), "GB": round(b / 1e9, 2), "calls": c,
                  "GB/s": round(b / s / 1e9, 1) if s else None}
    print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
