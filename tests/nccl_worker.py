"""Child process of tests/test_gpu_config4.py: a world of ONE rank on the RCCL backend
(torch.distributed "nccl"), so that the product's per-chunk collective -- communicator set-up,
``all_reduce(async_op=True)`` on RCCL's stream behind the chunk's kernel, ``work.wait()`` before the
epilogue -- executes on real hardware even though the test box has a single GPU.

    MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/nccl_worker.py OUT.npz NT NZ NY NX STEPS
"""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from momlevel_amd import core, parallel, synthetic  # noqa: E402


def main():
    out_path = sys.argv[1]
    nt, nz, ny, nx, steps = (int(v) for v in sys.argv[2:7])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1,
                            init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}")
    assert dist.get_backend() == "nccl"
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pres = np.asarray(g["z_l"]) * 1.0e4 + 101325.0
    kw = dict(seed=synthetic.SEED, mask3d=vol0)
    T = core.synth_field((nt, nz, ny, nx), field_id=1, lo=-2.0, scale=34.0, **kw)
    S = core.synth_field((nt, nz, ny, nx), field_id=2, lo=30.0, scale=10.0, **kw)
    variants = ("steric", "thermosteric", "halosteric")
    # the identity path (no collective) first, then the same walk through RCCL
    plain = parallel.steric_global_tile_streamed((T, S), vol0, g["areacello"], pres,
                                                 variants=variants, steps=steps, heat=True)
    ex = parallel.ChunkedExchange(1, force=True)
    assert ex.active, "the forced exchange must be active in a world of one"
    forced = parallel.steric_global_tile_streamed((T, S), vol0, g["areacello"], pres,
                                                  variants=variants, steps=steps, heat=True,
                                                  force_collective=True)
    # the labelled front end's exchange (a host vector through a device all-reduce) as well
    vec = parallel._sum_over_ranks()(np.arange(5.0))
    with open("/proc/self/maps") as f:
        maps = f.read()
    save = {"rccl_loaded": np.array("librccl" in maps or "libnccl" in maps),
            "host_vector": vec, "heat_plain": plain["heat"], "heat_forced": forced["heat"]}
    for v in variants:
        for k in ("masso", "eta", "volo", "masso0", "area_sum"):
            save[f"{v}_{k}_plain"] = np.asarray(plain[v][k])
            save[f"{v}_{k}_forced"] = np.asarray(forced[v][k])
    np.savez(out_path, **save)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
