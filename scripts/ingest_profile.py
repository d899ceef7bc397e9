"""The step before the path, end to end under rocprofv3: the public
steric() on HOST float32 inputs -- what MOM6 writes -- with roctx ranges around every chunk's staging,
kernels and downloads, so that the copy / kernel traces show what overlaps what.

    rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --output-format csv \
        -d gpurun_out/prof_r03_ingest -o run -- python3 scripts/ingest_profile.py
    python scripts/summarize_ingest.py gpurun_out/prof_r03_ingest profiles/r03_ingest_overlap.json
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOMLEVEL_AMD_ROCTX"] = "1"

import momlevel_amd as m  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ingest_check import dataset  # noqa: E402


def main():
    nt, nz, ny, nx = 96, 75, 576, 360  # 11.9 GB of float32 theta+S: three 4 GiB upload chunks
    d = dataset(nt, nz, ny, nx, np.float32)
    cells = nt * nz * ny * nx
    for domain in ("global", "local"):
        m.steric(d, domain=domain)  # warm-up: page-locked buffers are allocated here
        torch.cuda.synchronize()
        torch.cuda.nvtx.range_push(f"steric(domain={domain}) on host float32 inputs")
        t0 = time.perf_counter()
        res = m.steric(d, domain=domain)  # (kept: giving 12 GB of results back to the OS is not the call)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.nvtx.range_pop()
        del res
        print(f"{domain}: {dt * 1e3:.1f} ms, {cells / dt / 1e6:.1f} Mcells/s, "
              f"H2D {2 * cells * 4 / dt / 1e9:.1f} GB/s", flush=True)
    del d
    # the reference's one recorded real-size call (examples/example.ipynb cell 6), its own roctx range
    import example_call

    print(example_call.run(reps=3), flush=True)


if __name__ == "__main__":
    main()
