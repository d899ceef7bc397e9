"""Child process of tests/test_gpu_multirank.py::test_eight_ranks_two_by_four: ONE rank of an
8-rank world that never touches the GPU.  A one-GPU box admits at most 6 processes on its card, so
the 2x4 layout of BASELINE.json configs[3] cannot be rehearsed there with 8 HIP contexts; instead
the parent computed every tile's partial sums with the HIP kernels (one tile after the other) and
this rank carries ITS tile's partials through the product's exchange -- ChunkedExchange, one
collective per time chunk over gloo, then the replicated host epilogue -- exactly what
parallel.steric_global_tile_streamed does with them after K1.

    RANK=r WORLD_SIZE=8 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/rank_worker_exchange.py \
        PARTIALS_r.npz OUT_r.npz STEPS
"""

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from momlevel_amd import parallel  # noqa: E402


def main():
    part = dict(np.load(sys.argv[1]))
    out_path, steps = sys.argv[2], int(sys.argv[3])
    # NOT parallel.init_from_env(): it asks torch.cuda.is_available(), which opens the GPU (also
    # with HIP_VISIBLE_DEVICES empty) -- and 8 such ranks beside the parent were killed by the box's
    # process guard ("9 processes had the GPU open (limit 6)").  These ranks never make a HIP call
    # (and never call torch.distributed.barrier(), which opens the GPU too).
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.distributed.init_process_group(backend="gloo", rank=rank, world_size=world)
    rows = part["rows"]  # (nrows, nt): this tile's masso(t) per variant (+ heat)
    names = [str(n) for n in part["names"]]
    nt = rows.shape[1]
    ex = parallel.ChunkedExchange(rows.shape[0])
    assert ex.active
    for t0 in range(0, nt, steps):
        tail = (part["volo"], rows[0, 0], part["area"]) if t0 == 0 else None
        ex.add(torch.from_numpy(np.ascontiguousarray(rows[:, t0:t0 + steps])), tail)
    red, volo, masso0, area = ex.finish()
    save = {"rank": rank, "world": world}
    for i, name in enumerate(names):
        if name == "heat":
            save["heat"] = red[i].numpy()
            continue
        fin = parallel.finalize(red[i], volo, masso0, area)
        for k in ("masso", "eta", "volo", "masso0", "area_sum", "reference_height", "expansion_coeff"):
            save[f"{name}_{k}"] = np.asarray(fin[k])
    fds = []
    for f in os.listdir("/proc/self/fd"):
        try:
            fds.append(os.readlink(f"/proc/self/fd/{f}"))
        except OSError:
            pass
    save["gpu_open"] = np.array(any("/dev/kfd" in f or "/dev/dri" in f for f in fds))
    np.savez(out_path, **save)
    # (not torch.distributed.barrier(): it opens the GPU even on a gloo group -- measured on the box,
    #  scripts/diag/kfd_steps.py; an all-reduce of one host double is barrier enough)
    torch.distributed.all_reduce(torch.zeros(1, dtype=torch.float64))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
