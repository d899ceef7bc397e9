"""Developer smoke check on a GPU box: parity vs the oracle + a first bandwidth number.

    python scripts/gpu_devcheck.py [nt_big]

Not part of the product or of the test suite (tests/ holds the real parity tests).
"""

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from momlevel_amd import core, synthetic  # noqa: E402
from oracle import momlevel_numpy as o  # noqa: E402


def cmp(name, got, ref, exact=False):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = np.asarray(ref)
    nan_ok = np.array_equal(np.isnan(got), np.isnan(ref))
    m = ~np.isnan(ref)
    bit = np.array_equal(got[m], ref[m])
    rel = 0.0
    if m.any():
        denom = np.maximum(np.abs(ref[m]), 1e-300)
        rel = float(np.max(np.abs(got[m] - ref[m]) / denom))
    print(f"  {name:34s} nan-mask {'OK' if nan_ok else 'MISMATCH'}  bit-exact {bit}  max rel {rel:.3e}")
    return nan_ok and (bit or not exact)


def small_cases():
    print("== 5x5x5x5 reference test data (generic path: odd plane)")
    d = o.generate_test_data()
    dev = "cuda"
    T = torch.from_numpy(d["thetao"]).to(dev)
    S = torch.from_numpy(d["so"]).to(dev)
    pres = o.pressure_from_depth(d["z_l"])
    ref = o.setup_reference_state(d["thetao"], d["so"], d["volcello"], d["areacello"], d["z_l"])
    rho = core.eos_map(T, S, pres)
    cmp("rho (K0)", rho, o.calc_rho(d["thetao"], d["so"], pres), exact=True)
    vol0 = torch.from_numpy(ref["volcello"]).to(dev)
    masso = core.steric_global_masso(T, S, vol0, pres)
    cmp("masso (K1)", masso, o.calc_masso(o.calc_rho(d["thetao"], d["so"], pres), ref["volcello"]))
    res, _ = o.steric(d["thetao"], d["so"], d["volcello"], d["areacello"], d["z_l"], d["z_i"], d["deptho"])
    rho0m = core.fold_mask(torch.from_numpy(ref["rho"]).to(dev), vol0)
    drho, eta = core.steric_local(T, S, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=d["z_i"], deptho=d["deptho"])
    cmp("delta_rho (K2)", drho, res["delta_rho"], exact=True)
    cmp("eta (K2)", eta, res["steric"], exact=True)


def medium_case(ny=48, nx=64, nz=15, nt=11):
    print(f"== synthetic {nt}x{nz}x{ny}x{nx} with land mask (fast path)")
    g = synthetic.make_grid(ny, nx, nz)
    dev = "cuda"
    vol0 = torch.from_numpy(g["volcello"]).to(dev)
    kw = dict(seed=synthetic.SEED, mask3d=g["volcello"])
    Tn = synthetic.field_numpy((nt, nz, ny, nx), field_id=1, lo=-2.0, scale=34.0, **kw)
    Sn = synthetic.field_numpy((nt, nz, ny, nx), field_id=2, lo=30.0, scale=10.0, **kw)
    T = core.synth_field((nt, nz, ny, nx), seed=synthetic.SEED, field_id=1, lo=-2.0, scale=34.0, mask3d=vol0)
    S = core.synth_field((nt, nz, ny, nx), seed=synthetic.SEED, field_id=2, lo=30.0, scale=10.0, mask3d=vol0)
    cmp("synth theta replay", T, Tn, exact=True)
    cmp("synth so replay", S, Sn, exact=True)
    vol4 = np.broadcast_to(g["volcello"], Tn.shape).copy()
    for variant in ("steric", "thermosteric", "halosteric"):
        res, ref = o.steric(Tn, Sn, vol4, g["areacello"], g["z_l"], variant=variant, domain="global")
        Tv = T if variant != "halosteric" else T[0]
        Sv = S if variant != "thermosteric" else S[0]
        pres = o.pressure_from_depth(g["z_l"])
        masso = core.steric_global_masso(Tv, Sv, vol0, pres)
        cmp(f"masso {variant}", masso, res["masso"])
        res, ref = o.steric(Tn, Sn, vol4, g["areacello"], g["z_l"], g["z_i"], g["deptho"], variant=variant)
        rho0m = core.fold_mask(torch.from_numpy(ref["rho"]).to(dev), vol0)
        drho, eta = core.steric_local(Tv, Sv, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=g["z_i"], deptho=g["deptho"])
        cmp(f"delta_rho {variant}", drho, res["delta_rho"], exact=True)
        cmp(f"eta {variant}", eta, res[variant], exact=True)
    m0 = core.steric_global_masso(T[0:1], S[0:1], vol0, pres)
    m = core.steric_global_masso(T, S, vol0, pres)
    print("  masso(nt=1)[0] == masso(nt)[0] bitwise:", bool(m0[0] == m[0]))


def timing(ny, nx, nz, nt, reps=3):
    print(f"== timing K1 global fp64 {nt}x{nz}x{ny}x{nx}")
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    shape = (nt, nz, ny, nx)
    T = core.synth_field(shape, seed=synthetic.SEED, field_id=1, lo=-2.0, scale=34.0, mask3d=vol0)
    S = core.synth_field(shape, seed=synthetic.SEED, field_id=2, lo=30.0, scale=10.0, mask3d=vol0)
    pres = torch.from_numpy(o.pressure_from_depth(g["z_l"])).cuda()
    torch.cuda.synchronize()
    cells = nt * nz * ny * nx
    for name, fn in (
        ("K1 steric", lambda: core.steric_global_masso(T, S, vol0, pres)),
        ("K1 thermo", lambda: core.steric_global_masso(T, S[0], vol0, pres)),
    ):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        bpc = 16 if name.endswith("steric") else 8
        print(f"  {name}: {best*1e3:.3f} ms  {cells/best/1e6:.0f} Mcells/s  {bpc*cells/best/1e9:.0f} GB/s algorithmic")
    if nt <= 16:
        rho0 = core.eos_map(T[0], S[0], pres)
        rho0m = core.fold_mask(rho0, vol0)
        zi = torch.from_numpy(g["z_i"]).cuda()
        dep = torch.from_numpy(g["deptho"]).cuda()
        drho = torch.empty(shape, dtype=torch.float64, device="cuda")
        eta = torch.empty((nt, ny, nx), dtype=torch.float64, device="cuda")
        for name, want in (("K2 local +drho", True), ("K2 local eta only", False)):
            fn = lambda: core.steric_local(T, S, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=zi, deptho=dep,
                                           want_delta_rho=want, delta_rho_out=drho, eta_out=eta)
            fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e-3)
            bpc = 24 if want else 16
            print(f"  {name}: {best*1e3:.3f} ms  {cells/best/1e6:.0f} Mcells/s  {bpc*cells/best/1e9:.0f} GB/s algorithmic")


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info())
    small_cases()
    medium_case()
    timing(576, 360, 75, 12)
    nt_big = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    timing(1080, 1440, 75, nt_big)
