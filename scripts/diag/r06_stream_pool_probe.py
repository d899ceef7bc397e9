#!/usr/bin/env python3
"""Is one of torch's pooled HIP streams slow for device->host copies?  torch.cuda.Stream() hands out
streams from a pool of 32 (round robin); hostio makes three new ones per steric() call.  D2H of 1 GiB
(64 MiB pieces into page-locked buffers) on each of 40 consecutively created streams, alone and with
an H2D copy running on another stream."""
import json
import time

import torch

PIECE = 64 << 20
dev = torch.device("cuda", 0)
src = torch.empty(PIECE, dtype=torch.uint8, device=dev)
hout = [torch.empty(PIECE, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
hin = [torch.empty(PIECE, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
din = torch.empty(4 * PIECE, dtype=torch.uint8, device=dev)
up = torch.cuda.Stream(dev)
torch.cuda.synchronize()
rows = []
for k in range(40):
    s = torch.cuda.Stream(dev)
    res = {"k": k, "stream": hex(s.cuda_stream)}
    for mode in ("alone", "beside_h2d"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "beside_h2d":
            with torch.cuda.stream(up):
                for i in range(16):
                    din[(i % 4) * PIECE:(i % 4 + 1) * PIECE].copy_(hin[i % 4], non_blocking=True)
        with torch.cuda.stream(s):
            for i in range(16):
                hout[i % 4].copy_(src, non_blocking=True)
        s.synchronize()
        dt = time.perf_counter() - t0
        res[mode + "_GB/s"] = round(16 * PIECE / dt / 1e9, 1)
        torch.cuda.synchronize()
    rows.append(res)
slow = [r for r in rows if r["alone_GB/s"] < 40 or r["beside_h2d_GB/s"] < 35]
print(json.dumps({"slow_streams": slow, "distinct_streams": len({r["stream"] for r in rows}),
                  "alone_GB/s": [r["alone_GB/s"] for r in rows],
                  "beside_h2d_GB/s": [r["beside_h2d_GB/s"] for r in rows]}))
