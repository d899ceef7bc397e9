#!/usr/bin/env python3
"""Where inside the recorded call (scripts/example_call.py) does the wall time go BEFORE the first
result piece starts down the link and AFTER the last one arrives?  Timestamps (ms from the call's
start) of the phases of steric(): validation, reference state, first upload done, first download
submitted, last kernel enqueued, results complete."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import example_call  # noqa: E402
import momlevel_amd  # noqa: E402,F401
from momlevel_amd import engine, hostio  # noqa: E402

steric_mod = sys.modules["momlevel_amd.steric"]

marks = []
T0 = [0.0]


def mark(name):
    marks.append((name, round((time.perf_counter() - T0[0]) * 1e3, 1)))


def wrap(mod, name, before=None, after=None):
    fn = getattr(mod, name)

    def w(*a, **k):
        if before:
            mark(before)
        try:
            return fn(*a, **k)
        finally:
            if after:
                mark(after)
    setattr(mod, name, w)


wrap(steric_mod, "validate_dataset", "validate>", "validate<")
wrap(steric_mod, "_setup", "setup>", "setup<")
wrap(steric_mod, "_local_results", "local>", "local<")
wrap(engine, "local_steric_variants", "engine>", "engine<")
first = {"submit": True}
sub = hostio.Downloader.submit


def submit(self, pairs):
    if first["submit"]:
        first["submit"] = False
        mark("first download submitted")
    return sub(self, pairs)


hostio.Downloader.submit = submit
fin = hostio.Downloader.finish


def finish(self):
    mark("loop done, waiting for downloads")
    fin(self)
    mark("downloads complete")


hostio.Downloader.finish = finish
dinit = hostio.Downloader.__init__


def downloader_init(self, *a, **k):
    dinit(self, *a, **k)
    mark("downloader stream " + hex(self.stream.cuda_stream))


hostio.Downloader.__init__ = downloader_init
# where does a result piece spend its time: waiting for its DMA, or being copied out of the ring?
acc = {"drain_s": 0.0, "copy_out_s": 0.0, "copy_in_s": 0.0, "pieces": 0}
drain = hostio._drain_one
hcopy = hostio._host_copy
in_drain = threading.local()


def timed_copy(dst, src):
    t0 = time.perf_counter()
    hcopy(dst, src)
    acc["copy_out_s" if getattr(in_drain, "on", False) else "copy_in_s"] += time.perf_counter() - t0


def timed_drain(ring, pending):
    in_drain.on = True
    t0 = time.perf_counter()
    try:
        drain(ring, pending)
    finally:
        in_drain.on = False
        acc["drain_s"] += time.perf_counter() - t0
        acc["pieces"] += 1


hostio._host_copy = timed_copy
hostio._drain_one = timed_drain
tinit = engine.TimeChunks.__init__


def chunks_init(self, *a, **k):
    tinit(self, *a, **k)
    if self._copy_stream is not None:
        mark("upload stream " + hex(self._copy_stream.cuda_stream))


engine.TimeChunks.__init__ = chunks_init
runs = []


accs = []


def before():
    if marks:
        runs.append(list(marks))
        accs.append({k: round(v, 3) for k, v in acc.items()})
    for k in acc:
        acc[k] = 0
    marks.clear()
    first["submit"] = True
    T0[0] = time.perf_counter()


out = example_call.run(reps=int(os.environ.get("REPS", "5")), before_call=before)
runs.append(list(marks))
accs.append({k: round(v, 3) for k, v in acc.items()})
slow = [r for r, w in zip(runs, out["wall_s"]) if w > 1.0]
streams = [[m[0].split()[-1][-5:] for m in r if "stream" in m[0]] for r in runs]
print(json.dumps({"wall_s": out["wall_s"], "first_call_marks_ms": runs[0], "last_call_marks_ms": runs[-1], "slow_calls_marks_ms": slow, "streams_per_call": streams,
                  "download_pieces_per_call": accs}))
