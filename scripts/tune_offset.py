#!/usr/bin/env python3
"""Does the relative placement of the theta and S records in HBM matter to K1?  (Two interleaved
read streams could collide on channels/banks if their distance is a multiple of the interleave
period.)  Times the headline launch with S shifted by a few byte offsets inside one allocation.

    python scripts/tune_offset.py > profiles/r02_tune_stream_offset.log
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, synthetic  # noqa: E402


def timeit(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), float(np.mean(ts))


def main():
    nt, nz, ny, nx = 120, 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    dev = torch.device("cuda", 0)
    vol0 = torch.from_numpy(g["volcello"]).to(dev)
    pres = np.asarray(g["z_l"]) * 1.0e4 + 101325.0
    n = nt * nz * ny * nx
    pad = (64 << 20) // 8  # 64 MiB of slack in elements
    T = torch.empty(n, dtype=torch.float64, device=dev)
    Sraw = torch.empty(n + pad, dtype=torch.float64, device=dev)
    kw = dict(seed=synthetic.SEED, mask3d=vol0, device=dev)
    shape = (nt, nz, ny, nx)
    core.synth_field(shape, field_id=1, lo=-2.0, scale=34.0, out=T.view(shape), **kw)
    base = None
    print(f"# T at {T.data_ptr():#x}, S allocation at {Sraw.data_ptr():#x}; cells {n:.4e}")
    for off_bytes in (0, 256, 1024, 4096, 65536, 1 << 20, (1 << 20) + 4096, (16 << 20) + 2048,
                      (32 << 20) + 256 * 37):
        off = off_bytes // 8
        S = Sraw[off:off + n].view(shape)
        core.synth_field(shape, field_id=2, lo=30.0, scale=10.0, out=S, **kw)
        Tv = T.view(shape)
        best, mean = timeit(lambda: core.steric_global_masso(Tv, S, vol0, pres, skip_dry=False))
        m = core.steric_global_masso(Tv, S, vol0, pres, skip_dry=False).cpu().numpy()
        if base is None:
            base = m
        assert np.array_equal(m, base)
        dist = S.data_ptr() - T.data_ptr()
        print(f"S offset {off_bytes:>10d} B  (S - T = {dist:#x})  best {best:7.3f} ms  mean {mean:7.3f} ms"
              f"  {16 * n / best / 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
