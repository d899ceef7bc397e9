"""Child process of tests/test_gpu_multirank.py: one rank of momlevel_amd.parallel.steric (the
labelled, reference-signature API on a horizontal tile).

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p MOMLEVEL_AMD_DIST_BACKEND=gloo \
        python tests/rank_worker_labelled.py OUT.npz NT NZ NY NX DTYPE
"""

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from momlevel_amd import parallel, synthetic  # noqa: E402
from momlevel_amd.labeled import DataArray, Dataset  # noqa: E402


def tile_dataset(nt, nz, ny, nx, dtype, rank, world):
    y0, y1, x0, x1 = synthetic.tile_bounds(ny, nx, rank, world)
    g = synthetic.make_grid(ny, nx, nz)
    r = np.random.default_rng(7)
    mask = np.isnan(g["volcello"])
    T = np.where(mask[None], np.nan, r.normal(12.0, 6.0, (nt, nz, ny, nx))).astype(dtype)
    S = np.where(mask[None], np.nan, r.normal(35.0, 1.0, (nt, nz, ny, nx))).astype(dtype)
    d = Dataset()
    d["time"] = DataArray(np.arange(nt, dtype=float), ("time",))
    d["z_l"] = DataArray(g["z_l"], ("z_l",))
    d["z_i"] = DataArray(g["z_i"], ("z_i",))
    dims = ("time", "z_l", "yh", "xh")
    cut = (slice(None), slice(None), slice(y0, y1), slice(x0, x1))
    d["thetao"] = DataArray(np.ascontiguousarray(T[cut]), dims)
    d["so"] = DataArray(np.ascontiguousarray(S[cut]), dims)
    d["volcello"] = DataArray(np.ascontiguousarray(np.broadcast_to(g["volcello"], T.shape)[cut]), dims)
    d["areacello"] = DataArray(np.ascontiguousarray(g["areacello"][y0:y1, x0:x1]), ("yh", "xh"))
    d["deptho"] = DataArray(np.ascontiguousarray(g["deptho"][y0:y1, x0:x1]), ("yh", "xh"))
    return d, (y0, y1, x0, x1)


def main():
    out_path = sys.argv[1]
    nt, nz, ny, nx = (int(v) for v in sys.argv[2:6])
    dtype = np.float32 if sys.argv[6] == "f32" else np.float64
    rank, world, _ = parallel.init_from_env()
    d, bounds = tile_dataset(nt, nz, ny, nx, dtype, rank, world)
    save = {"bounds": np.array(bounds)}
    for variant in ("steric", "thermosteric"):
        res, ref = parallel.steric(d, variant=variant, domain="global")
        save[f"global_{variant}"] = res[variant].values
        save[f"global_{variant}_href"] = res["reference_height"].values
        for k in ("volo", "masso", "rhoga"):
            save[f"global_{variant}_{k}"] = ref[k].values
    res, ref = parallel.steric(d, domain="local")
    save["local_eta"] = res["steric"].values
    save["local_ref_masso"] = ref["masso"].values
    results, ref = parallel.steric_variants(d, domain="global", heat_content=True)
    for v in ("steric", "thermosteric", "halosteric"):
        save[f"variants_{v}"] = results[v][v].values
    save["variants_ohc"] = results["heat"]["ohc"].values
    sref = parallel.setup_reference_state(d)
    save["setup_masso"], save["setup_volo"] = sref["masso"].values, sref["volo"].values
    # a tile alone fails the areacello range check; the tiled API checks the global sum
    import momlevel_amd

    try:
        momlevel_amd.steric(d, domain="global")
        save["untiled_call_raises"] = np.array(False)
    except ValueError:
        save["untiled_call_raises"] = np.array(True)
    np.savez(out_path, **save)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
