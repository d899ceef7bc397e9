#!/usr/bin/env python3
"""Does the placement of the delta_rho output relative to the streamed input matter to the local
pass?  Round 4's A/B of K2's time steps per thread saw IDENTICAL held-field kernels with delta_rho
differ by up to 7 % between processes (14.9 ... 16.1 ms, profiles/r04_tune_k2_nti.log) -- the
buffers sat at different addresses.  Times the float32 and float64 thermosteric passes with
delta_rho, the output shifted by byte offsets inside one allocation (the input stays put).

    python scripts/tune_k2_offset.py > profiles/r04_tune_k2_output_offset.log
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, synthetic  # noqa: E402


def timeit(fn, reps=5):
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
    nt, nz, ny, nx = 48, 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    dev = torch.device("cuda", 0)
    vol0 = torch.from_numpy(g["volcello"]).to(dev)
    pz = torch.from_numpy(np.asarray(g["z_l"]) * 1.0e4 + 101325.0).to(dev)
    zi, dep = torch.from_numpy(g["z_i"]).to(dev), torch.from_numpy(g["deptho"]).to(dev)
    shape = (nt, nz, ny, nx)
    n = nt * nz * ny * nx
    pad = (160 << 20) // 8
    raw = torch.empty(n + pad, dtype=torch.float64, device=dev)
    eta = torch.empty((nt, ny, nx), dtype=torch.float64, device=dev)
    offsets = (0, 64, 256, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 20, (1 << 20) + 4096,
               (2 << 20), (16 << 20) + 2048, (32 << 20) + 256 * 37, (128 << 20))
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        kw = dict(seed=synthetic.SEED, mask3d=vol0, device=dev)
        T = core.synth_field(shape, dt, field_id=1, lo=-2.0, scale=34.0, **kw)
        S = core.synth_field(shape, dt, field_id=2, lo=30.0, scale=10.0, **kw)
        rho0m = core.fold_mask(core.eos_map(T[0], S[0], pz), vol0)
        print(f"# {tag}: theta at {T.data_ptr():#x}, output allocation at {raw.data_ptr():#x}, "
              f"{nt} steps, thermosteric + delta_rho and steric + delta_rho", flush=True)
        for off_bytes in offsets:
            out = raw[off_bytes // 8:off_bytes // 8 + n].view(shape)
            res = []
            for Sv in (S[0], S):
                best, mean = timeit(lambda: core.steric_local(
                    T, Sv, rho0m, vol0[0], pz, -1.0 / 1035.0, z_i=zi, deptho=dep,
                    delta_rho_out=out, eta_out=eta, skip_dry=False))
                res.append((best, mean))
            dist = out.data_ptr() - T.data_ptr()
            print(f"{tag} output offset {off_bytes:>10d} B (out - theta = {dist:#x} = {dist % (1 << 21):#x} "
                  f"mod 2 MiB)  thermo best {res[0][0]:7.3f} mean {res[0][1]:7.3f}   steric best "
                  f"{res[1][0]:7.3f} mean {res[1][1]:7.3f} ms", flush=True)
        del T, S, rho0m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
