"""Which step of a GPU-less rank process opens the GPU?  (prints /dev/kfd|dri fds after each)"""
import os, sys
def fds(tag):
    out = []
    for f in os.listdir("/proc/self/fd"):
        try: out.append(os.readlink(f"/proc/self/fd/{f}"))
        except OSError: pass
    print(tag, [f for f in out if "kfd" in f or "/dri" in f], flush=True)
fds("start")
import numpy as np; fds("numpy")
import torch; fds("import torch")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from momlevel_amd import parallel; fds("import momlevel_amd.parallel")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
torch.distributed.init_process_group(backend="gloo", rank=0, world_size=1); fds("init gloo")
ex = parallel.ChunkedExchange(1, force=True); fds("ChunkedExchange")
ex.add(torch.ones(1, 3, dtype=torch.float64), (1.0, 2.0, 3.0)); fds("add")
r = ex.finish(); fds("finish")
parallel.finalize(r[0][0], r[1], r[2], r[3]); fds("finalize")
torch.distributed.all_reduce(torch.zeros(1, dtype=torch.float64)); fds("all_reduce as barrier")
torch.distributed.barrier(); fds("barrier")
print("device_count", torch.cuda.device_count()); fds("device_count")
