#!/usr/bin/env python3
"""Does the NUMA node a page-locked staging buffer lives on decide the DMA rate into it?  The GPU box
gives this process a CPU QUOTA (16 CPUs' worth), not a cpuset: its threads may run on either socket,
and a buffer is placed where the thread that allocates (pins) it happens to run."""
import glob
import json
import os
import time

import torch

PIECE = 64 << 20


def cpus_of(node):
    out = []
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
allowed = sorted(os.sched_getaffinity(0))
info = {"nodes": nodes, "allowed_cpus": len(allowed),
        "gpu_numa_node": [open(p).read().strip() for p in glob.glob("/sys/class/drm/card*/device/numa_node")],
        "numa_balancing": open("/proc/sys/kernel/numa_balancing").read().strip()
        if os.path.exists("/proc/sys/kernel/numa_balancing") else None}
dev = torch.device("cuda", 0)
props = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                             getattr(props, "pci_device_id", 0))
info["gpu_bdf"] = bdf
try:
    info["this_gpu_numa_node"] = open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip()
except OSError as e:
    info["this_gpu_numa_node"] = str(e)
src = torch.empty(PIECE, dtype=torch.uint8, device=dev)
dst = torch.empty(PIECE, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
rows = []
for rep in range(3):
    for node in nodes:
        cpus = [c for c in cpus_of(node) if c in allowed]
        if not cpus:
            continue
        os.sched_setaffinity(0, cpus)
        time.sleep(0.01)
        # a size of its own per allocation: torch's pinned cache must not hand an old block out
        size = PIECE + 4096 * (1 + len(rows))
        bufs = [torch.empty(size, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
        for b in bufs:
            b[:PIECE].fill_(1)
        os.sched_setaffinity(0, allowed)
        res = {"buffers_on_node": node}
        for b in bufs:  # (warm: the first DMA into a freshly pinned buffer is slow whatever its node)
            b[:PIECE].copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        for name, fn in (("d2h", lambda b: b[:PIECE].copy_(src, non_blocking=True)),
                         ("h2d", lambda b: dst.copy_(b[:PIECE], non_blocking=True))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(32):
                fn(bufs[i % 4])
            torch.cuda.synchronize()
            res[name + "_GB/s"] = round(32 * PIECE / (time.perf_counter() - t0) / 1e9, 1)
        rows.append(res)
        del bufs
info["rates"] = rows
print(json.dumps(info))
