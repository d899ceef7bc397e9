#!/usr/bin/env python3
"""Repeated host-input calls in one process: does anything grow?  Resident set size, live threads
and the wall time of every call of the reference's recorded example (scripts/example_call.py,
smaller by default) -- a leak of worker threads, staging rings or page-locked blocks would show.

    python scripts/host_path_soak.py [--calls 12] [--nt 24]
"""
import argparse
import json
import os
import sys
import threading

import psutil

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import example_call  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=12)
    ap.add_argument("--nt", type=int, default=24)
    a = ap.parse_args()
    proc = psutil.Process()
    rows = []

    def before():
        rows.append({"rss_GB": round(proc.memory_info().rss / 1e9, 2), "threads": threading.active_count(),
                     "os_threads": proc.num_threads()})

    out = example_call.run(a.nt, 35, reps=a.calls, before_call=before)
    before()
    print(json.dumps({"nt": a.nt, "wall_s": out["wall_s"], "before_each_call_and_after_the_last": rows}))


if __name__ == "__main__":
    main()
