#!/usr/bin/env python3
"""The host link's floor for a call that moves `--in-gb` to the device and `--out-gb` back AT THE
SAME TIME: page-locked buffers of our own on the host side (no host copies at all, no fresh pages),
64 MiB pieces, one stream per direction, everything enqueued up front.  What is left between this
number and the wall time of the reference's recorded call (scripts/example_call.py: 13.06 GB in,
26.87 GB out) is the product's host work -- staging copies, first touch of result pages, pipeline
fill and drain -- and nothing else.

    python scripts/link_duplex_probe.py [--in-gb 13.06] [--out-gb 26.87] [--reps 4]
"""
import argparse
import json
import time

import torch

PIECE = 64 << 20


def run(in_gb, out_gb, reps):
    dev = torch.device("cuda", 0)
    n_in, n_out = int(in_gb * 1e9) // PIECE, int(out_gb * 1e9) // PIECE
    hin = [torch.empty(PIECE, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
    hout = [torch.empty(PIECE, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
    din = torch.empty(4 * PIECE, dtype=torch.uint8, device=dev)
    dout = torch.empty(4 * PIECE, dtype=torch.uint8, device=dev)
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()

    def go(do_in, do_out):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        if do_in:
            with torch.cuda.stream(s_in):
                ev[0].record()
                for k in range(n_in):
                    din[(k % 4) * PIECE:(k % 4 + 1) * PIECE].copy_(hin[k % 4], non_blocking=True)
                ev[1].record()
        if do_out:
            with torch.cuda.stream(s_out):
                ev[2].record()
                for k in range(n_out):
                    hout[k % 4].copy_(dout[(k % 4) * PIECE:(k % 4 + 1) * PIECE], non_blocking=True)
                ev[3].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        t_in = ev[0].elapsed_time(ev[1]) * 1e-3 if do_in else None
        t_out = ev[2].elapsed_time(ev[3]) * 1e-3 if do_out else None
        return wall, t_in, t_out

    out = {"in_GB": round(n_in * PIECE / 1e9, 2), "out_GB": round(n_out * PIECE / 1e9, 2),
           "piece_MiB": PIECE >> 20}
    for name, (a, b) in (("h2d_alone", (True, False)), ("d2h_alone", (False, True)),
                         ("both_at_once", (True, True))):
        walls, rin, rout = [], [], []
        for _ in range(reps):
            w, ti, to = go(a, b)
            walls.append(round(w, 4))
            if ti:
                rin.append(round(n_in * PIECE / ti / 1e9, 1))
            if to:
                rout.append(round(n_out * PIECE / to / 1e9, 1))
        out[name] = {"wall_s": walls, "h2d_GB/s": rin, "d2h_GB/s": rout}
    out["floor_s_for_this_byte_mix"] = min(out["both_at_once"]["wall_s"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-gb", type=float, default=13.06)
    ap.add_argument("--out-gb", type=float, default=26.87)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    print(json.dumps(run(a.in_gb, a.out_gb, a.reps)), flush=True)


if __name__ == "__main__":
    main()
