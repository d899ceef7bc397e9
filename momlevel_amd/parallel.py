"""Multi-GPU global steric: horizontal tiles + ONE all-reduce.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The
(yh, xh) plane is tiled across ranks (1x2, 2x2, 2x4); every rank runs K1 on its
own tile for all z and t -- the tiles are independent, nothing is exchanged on
the data path -- and the only collective is a single ``all_reduce(SUM)`` of the
packed float64 vector

    [ masso(t) for every local time step | volo | masso0 | sum(areacello) ]

((nt+3)*8 bytes: latency-bound, a few tens of microseconds on xGMI).  Every rank
then evaluates ``h_ref * ln(rhoga0 * volo / masso(t))`` redundantly.  masso0 and
masso(t) travel in the same vector and are summed in the same rank order, so
``steric[t=0] == 0`` holds exactly for any world size, as on one GPU.

The local variants need no collective at all (columns are independent): each
rank simply runs ``engine.local_steric`` on its tile.
"""

import os

import numpy as np
import torch
import torch.distributed as dist

from . import engine


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

    Returns (rank, world_size, local_rank).  A single-process run (no WORLD_SIZE or
    WORLD_SIZE=1) does not create a process group.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MOMLEVEL_AMD_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer
            # GPUs than ranks (device tensors are staged through the host for the exchange)
            backend = os.environ.get("MOMLEVEL_AMD_DIST_BACKEND") or (
                "nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():  # every backend: the rank's kernels go to ITS GPU
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def pack_partials(masso, volo, masso0, area_sum):
    """-> one float64 vector [masso(0..nt-1), volo, masso0, area_sum] on masso's device."""
    tail = torch.stack([
        torch.as_tensor(v, dtype=torch.float64, device=masso.device).reshape(())
        for v in (volo, masso0, area_sum)
    ])
    return torch.cat([masso.to(torch.float64).reshape(-1), tail])


def exchange_global(masso, volo, masso0, area_sum, group=None):
    """The path's single exchange step: all-reduce the packed partial sums.

    Works on device tensors with RCCL ("nccl") and on CPU tensors with gloo.
    Without an initialised process group (single GPU) it is the identity.
    Returns (masso (nt,), volo, masso0, area_sum) as tensors on the input device.
    """
    vec = pack_partials(masso, volo, masso0, area_sum)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if vec.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal path, see init_from_env
            host = vec.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            vec = host.to(vec.device)
        else:
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    nt = vec.numel() - 3
    return vec[:nt], vec[nt], vec[nt + 1], vec[nt + 2]


def finalize(masso, volo, masso0, area_sum):
    """Host epilogue on the all-reduced sums (steric.py:136-142, derived.py:662)."""
    masso = masso.detach().cpu().numpy()
    volo, masso0, area_sum = (np.float64(x.item()) for x in (volo, masso0, area_sum))
    rhoga = masso0 / volo
    reference_height, eta, expansion = engine.global_finalize(masso, volo, rhoga, area_sum)
    return {
        "masso": masso,
        "volo": volo,
        "masso0": masso0,
        "rhoga": rhoga,
        "area_sum": area_sum,
        "reference_height": reference_height,
        "expansion_coeff": expansion,
        "eta": eta,
    }


def steric_global_tile(T, S, vol0, areacello, pres, variant="steric", eos="wright",
                       f32_mode="faithful", group=None, validate_area=True):
    """Global steric of a horizontally tiled grid; call on every rank with its tile.

    T, S: (nt,nz,ny_t,nx_t) device tensors of this rank's tile; vol0 (nz,ny_t,nx_t) the
    reference volcello (time index 0); areacello (ny_t,nx_t).  The reference state is
    time index 0 of (T, S), as setup_reference_state builds it.
    """
    from . import core

    if variant == "thermosteric":
        Tv, Sv = T, S[0]
    elif variant == "halosteric":
        Tv, Sv = T[0], S
    elif variant == "steric":
        Tv, Sv = T, S
    else:
        raise ValueError(f"Unknown variant '{variant}' passed to `steric`")
    _rho0, volo, _ = engine.reference_state(T[0], S[0], vol0, pres, eos=eos, f32_mode=f32_mode,
                                            with_masso=False, with_rho=False)
    masso = engine.global_masso(Tv, Sv, vol0, pres, eos=eos, f32_mode=f32_mode)
    masso0 = masso[0]  # the reference slab is step 0 of this record: same launch, same bits
    area = core.nansum(engine.to_device(areacello, masso.device, torch.float64))
    masso, volo, masso0, area = exchange_global(masso, volo, masso0, area, group=group)
    out = finalize(masso, volo, masso0, area)
    if validate_area:  # util.validate_areacello on the GLOBAL sum, not the tile's
        err = (out["area_sum"] - 3.6111092e14) / 3.6111092e14
        if not abs(err) < 0.02:
            raise ValueError("Errors found in dataset.")
    return out


def steric_local_tile(T, S, vol0, pres, z_i, deptho, rhozero=1035.0, variant="steric",
                      eos="wright", f32_mode="faithful", want_delta_rho=True):
    """Local steric of a horizontally tiled grid: columns are independent, so a rank simply runs
    the fused K2 pass on its own (nt,nz,ny_t,nx_t) tile -- NO collective on this path (SURVEY.md
    8e); gathering the (time,yh,xh) tiles, if wanted at all, is an output step of the caller.

    Returns (delta_rho or None, eta) for the tile; the reference state is time index 0.
    """
    from . import core

    if variant == "thermosteric":
        Tv, Sv = T, S[0]
    elif variant == "halosteric":
        Tv, Sv = T[0], S
    elif variant == "steric":
        Tv, Sv = T, S
    else:
        raise ValueError(f"Unknown variant '{variant}' passed to `steric`")
    rho0 = core.eos_map(T[0], S[0], pres, eos=eos, f32_mode=f32_mode)
    return engine.local_steric(Tv, Sv, rho0, vol0, pres, rhozero, z_i=z_i, deptho=deptho, eos=eos,
                               f32_mode=f32_mode, want_delta_rho=want_delta_rho, out_host=False)
