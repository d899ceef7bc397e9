#!/usr/bin/env python3
"""K1 steric (the headline kernel), fused arithmetic, nt=120 float64: time chunks 24..64 interleaved over
several rounds in one process -- is 32 still the chunk to run?  (MLX_FLAG_TCHUNK never changes a bit.)

    python scripts/tune_tchunk_steric.py [--rounds 6]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    nt, nz, ny, nx = 120, 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    dev = torch.device("cuda", 0)
    vol0 = torch.from_numpy(g["volcello"]).to(dev)
    pres = np.asarray(g["z_l"]) * 1.0e4 + 101325.0
    kw = dict(seed=synthetic.SEED, mask3d=vol0, device=dev)
    T = core.synth_field((nt, nz, ny, nx), torch.float64, field_id=1, lo=-2.0, scale=34.0, **kw)
    S = core.synth_field((nt, nz, ny, nx), torch.float64, field_id=2, lo=30.0, scale=10.0, **kw)
    chunks = [24, 32, 40, 48, 56, 64]
    ref = core.steric_global_masso(T, S, vol0, pres, skip_dry=False, t_chunk=32)
    times = {tc: [] for tc in chunks}
    for _ in range(a.rounds):
        for tc in chunks:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = core.steric_global_masso(T, S, vol0, pres, skip_dry=False, t_chunk=tc)
            e1.record()
            torch.cuda.synchronize()
            times[tc].append(e0.elapsed_time(e1))
            assert torch.equal(out, ref)
    gb = 16.0 * nt * nz * ny * nx / 1e9
    for tc in chunks:
        v = sorted(times[tc])
        print(f"t_chunk {tc:3d}: best {v[0]:.3f} ms  median {v[len(v) // 2]:.3f}  worst {v[-1]:.3f}   "
              f"{gb / v[len(v) // 2] / 8:.4f} of 8 TB/s at the median", flush=True)


if __name__ == "__main__":
    main()
