"""Multi-GPU global steric: horizontal tiles + one all-reduce per time chunk.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The
(yh, xh) plane is tiled across ranks (1x2, 2x2, 2x4); every rank runs K1 on its
own tile for all z and t -- the tiles are independent, nothing is exchanged on
the data path -- and the only collective is the sum over the ranks of the packed
float64 vector

    [ masso(t) for every time step of the chunk | volo | masso0 | sum(areacello) ]

once per time chunk (the last three ride in the first chunk's; (nt_chunk+3)*8 bytes:
latency-bound, a few tens of microseconds on xGMI, asynchronous and overlapped with the next
chunk's kernels).  A record that fits in HBM can go as ONE chunk; a 1200-step record
(BASELINE.json configs[3]) is walked in chunks, from resident tensors, host arrays or a
generator (``steric_global_tile_streamed``).  Every rank
then evaluates ``h_ref * ln(rhoga0 * volo / masso(t))`` redundantly.

The sum is RANK-ORDERED (``rank_ordered_sum``): one RCCL collective moves every rank's vector to
every rank (``all_gather_into_tensor``: world x (nt_chunk+3) doubles, still latency-bound) and each
rank adds the rows 0, 1, ..., N-1 in that order in float64.  Every ELEMENT of the vector is thereby
reduced in the same order -- which a library all-reduce does not promise: a ring reduces segment k
of the vector starting at rank k, so masso0 and masso(t=0) -- the same per-rank partials at two
positions of the vector -- came out 1 ulp apart in the round-5 rehearsal of the 8-rank 2x4 layout
over gloo (``steric[t=0] = 1.7e-15`` instead of 0; 2 ranks cannot show it, a + b is commutative).
With the ordered sum ``steric[t=0] == 0`` holds exactly for any world size, as on one GPU and in
the reference, for every variant; the result is bit-identical on all ranks and from run to run.
The host sum does not depend on the backend, so the gloo rehearsals compute what RCCL computes
provided its all-gather delivers every rank's vector intact -- UNVERIFIED at N>1: no multi-GPU run
of this code exists (in a world of one rank the forced RCCL all-gather is bit-identical to the
collective-free walk: tests/nccl_worker.py, bench.py --force-collective).  ``MOMLEVEL_AMD_EXCHANGE=
allreduce`` selects the library's own ``all_reduce(SUM)`` instead (same bytes on the wire per rank
up to the factor N; results within 1 ulp per element of the ordered sum, no exact-zero guarantee).

The local variants need no collective at all (columns are independent): each
rank simply runs ``engine.local_steric`` on its tile.
"""

import os

import numpy as np
import torch
import torch.distributed as dist

from . import engine
from .adapters import accepts_xarray


def choose_backend(gpus, local_world, requested=None):
    """The torch.distributed backend of a multi-rank run on this node: what was asked for
    (MOMLEVEL_AMD_DIST_BACKEND or the caller), else "nccl" (= RCCL) when every rank of the node has a
    GPU of its own, else "gloo" -- a rehearsal: RCCL cannot put two ranks on one GPU ("Duplicate GPU
    detected"), so ranks that share cards exchange through the host (bench.py says so in its line)."""
    if requested:
        return requested
    return "nccl" if gpus and local_world <= gpus else "gloo"


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
        # device_count() does not initialise the GPU (is_available() does, and opens the device even
        # when HIP_VISIBLE_DEVICES hides it): a rank that sees no GPU never touches one here -- the
        # CPU rehearsals of 8 ranks must not put 8 processes on a card that admits 6
        gpus = torch.cuda.device_count()
        # MOMLEVEL_AMD_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than
        # ranks (device tensors are staged through the host for the exchange); so does, by itself, a
        # launch of more ranks per node than the node has GPUs (torchrun sets LOCAL_WORLD_SIZE)
        backend = choose_backend(gpus, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))),
                                 backend or os.environ.get("MOMLEVEL_AMD_DIST_BACKEND"))
        if gpus:  # every backend: the rank's kernels go to ITS GPU
            torch.cuda.set_device(local_rank % gpus)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def host_barrier(group=None):
    """A barrier that never touches the GPU: ``torch.distributed.barrier()`` initialises the device
    even on a gloo group (measured on the MI355X box, scripts/diag/kfd_steps.py); an all-reduce of
    one host double over a gloo group does not.  For ranks that must stay off the card."""
    if dist.get_backend(group) == "gloo":
        dist.all_reduce(torch.zeros(1, dtype=torch.float64), group=group)
    else:
        dist.barrier(group=group)


def rank_environments(n, environ=None, port=None, visible_gpus=None):
    """The environments of ``n`` ranks of one node, as torch.distributed.run would set them:
    RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR (127.0.0.1: the container
    hostname may not resolve) / MASTER_PORT (a free port unless given).  With fewer visible GPUs than
    ranks (``visible_gpus``; a rehearsal box) and no MOMLEVEL_AMD_DIST_BACKEND set, the ranks are
    told to use gloo -- RCCL cannot put two ranks on one GPU."""
    import socket

    base = dict(os.environ if environ is None else environ)
    if port is None:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port))
    if (visible_gpus is not None and visible_gpus < n
            and not base.get("MOMLEVEL_AMD_DIST_BACKEND")):
        base["MOMLEVEL_AMD_DIST_BACKEND"] = "gloo"
    return [dict(base, RANK=str(r), LOCAL_RANK=str(r)) for r in range(n)]


def launch_local_ranks(n, argv, environ=None, visible_gpus=None, out=None, grace=30.0):
    """Run ``argv`` as ``n`` fresh rank processes on this node and wait for them: rank 0's stdout is
    relayed line by line to ``out`` (default sys.stdout), the other ranks' stdout goes to stderr.
    Returns the worst exit code; when a rank fails, the survivors get ``grace`` seconds and are then
    terminated (exactly the processes started here).  The CALLER must not have touched the GPU:
    the ranks are children of this process (``python bench.py --gpus N`` without torchrun)."""
    import subprocess
    import sys
    import threading
    import time

    out = sys.stdout if out is None else out
    procs = []
    try:
        for env in rank_environments(n, environ, visible_gpus=visible_gpus):
            first = env["RANK"] == "0"
            # (file descriptor 2 itself: sys.stderr may be a capturing wrapper without one)
            procs.append(subprocess.Popen(list(argv), env=env,
                                          stdout=subprocess.PIPE if first else 2,
                                          text=first or None))
    except BaseException:
        for p in procs:  # never leave half a job behind
            p.kill()
        raise

    def relay():
        for line in procs[0].stdout:
            out.write(line)
            out.flush()

    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    deadline, killing = None, False
    while any(p.poll() is None for p in procs):
        if deadline is None and any(p.poll() not in (None, 0) for p in procs):
            deadline = time.time() + grace  # a rank died: the others are stuck in a collective
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    (p.kill if killing else p.terminate)()  # SIGTERM first, SIGKILL 10 s later
            deadline, killing = (float("inf") if killing else time.time() + 10.0), True
        time.sleep(0.05)
    pump.join(timeout=10)
    codes = [p.returncode for p in procs]
    return max((abs(c) for c in codes), default=0)


def exchange_mode():
    """"ordered" (default: all-gather + rank-ordered float64 sum) or "allreduce" (the library's
    all_reduce(SUM)); see the module docstring."""
    mode = os.environ.get("MOMLEVEL_AMD_EXCHANGE", "ordered")
    if mode not in ("ordered", "allreduce"):
        raise ValueError(f"MOMLEVEL_AMD_EXCHANGE={mode!r}: 'ordered' or 'allreduce'")
    return mode


def _in_a_world(group=None, force=False):
    return (dist.is_available() and dist.is_initialized()
            and (dist.get_world_size(group) > 1 or force))


def rank_ordered_sum(gathered):
    """(world, n) float64 numpy array of every rank's vector -> their sum, adding the ranks'
    rows in rank order 0, 1, ..., world-1 (explicitly: numpy's own reduction order over a leading
    axis is an implementation detail).  Host arithmetic on a few hundred doubles: part of the
    replicated host epilogue, like the logarithm that follows it."""
    gathered = np.asarray(gathered, dtype=np.float64)
    acc = gathered[0].copy()
    for r in range(1, gathered.shape[0]):
        acc += gathered[r]
    return acc


# What the collectives of this process have been, for logs and for the tests that must SEE a
# collective run (a forced exchange in a world of one is numerically the identity): counters only,
# nothing here feeds a result.
exchange_stats = {"collectives": 0, "on_device": 0, "doubles": 0, "last": None}


class _Exchange:
    """One sum-over-ranks of a packed float64 vector, started asynchronously.  Construction enqueues
    the collective (RCCL: on its own stream behind the kernels that produced ``vec``; gloo with a device
    vector -- a rehearsal on fewer GPUs than ranks -- stages through the host and is synchronous);
    ``result()`` waits and returns the sum as a float64 numpy vector on the host."""

    def __init__(self, vec, group=None):
        self.mode = exchange_mode()
        self.work = None
        world = dist.get_world_size(group)
        backend = dist.get_backend(group)
        if vec.is_cuda and backend == "gloo":
            vec = vec.cpu()
        vec = vec.contiguous()
        self._vec = vec  # (the collective reads it asynchronously: it stays alive until result())
        if self.mode == "allreduce":
            self.buf = vec.clone()
            self.work = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
        else:
            # (flat: rank r's vector lands at [r*n, (r+1)*n) -- the layout gloo and RCCL share)
            self.buf = torch.empty(world * vec.numel(), dtype=vec.dtype, device=vec.device)
            self.work = dist.all_gather_into_tensor(self.buf, vec.reshape(-1), group=group,
                                                    async_op=True)
        self.world = world
        exchange_stats["collectives"] += 1
        exchange_stats["on_device"] += int(self.buf.is_cuda)
        exchange_stats["doubles"] += vec.numel()
        exchange_stats["last"] = {"backend": backend, "mode": self.mode, "world": world,
                                  "device": str(self.buf.device), "doubles": vec.numel()}

    def result(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        host = self.buf.cpu().numpy()  # (a device buffer: synchronises with the collective)
        self._vec = None
        return host if self.mode == "allreduce" else rank_ordered_sum(host.reshape(self.world, -1))


def pack_partials(masso, volo, masso0, area_sum):
    """-> one float64 vector [masso(0..nt-1), volo, masso0, area_sum] on masso's device."""
    tail = torch.stack([
        torch.as_tensor(v, dtype=torch.float64, device=masso.device).reshape(())
        for v in (volo, masso0, area_sum)
    ])
    return torch.cat([masso.to(torch.float64).reshape(-1), tail])


def exchange_global(masso, volo, masso0, area_sum, group=None):
    """The path's single exchange step: the packed partial sums, summed over the ranks in rank
    order (module docstring).

    Works on device tensors with RCCL ("nccl") and on CPU tensors with gloo.
    Without an initialised process group (single GPU) it is the identity (nothing is copied or
    synchronised: the inputs come back as they are); in a world of several ranks the sums come
    back as float64 HOST tensors -- ``finalize`` takes either.
    Returns (masso (nt,), volo, masso0, area_sum).
    """
    vec = pack_partials(masso, volo, masso0, area_sum)
    if _in_a_world(group):
        vec = torch.from_numpy(_Exchange(vec, group).result())
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


_VARIANTS = ("steric", "thermosteric", "halosteric")


class ChunkedExchange:
    """The data-path collective of the tiled global variants, one exchange PER TIME CHUNK
    (SURVEY.md 8e): chunk k contributes the packed float64 vector

        [ masso_v(t) for every requested row v and every step t of the chunk ]
        ++ [ volo, masso0, sum(areacello) ]            (first chunk only)

    (``(rows*nt_chunk + 3) * 8`` bytes: latency-bound), summed over the ranks in rank order
    (module docstring: one all-gather + an ordered host sum; MOMLEVEL_AMD_EXCHANGE=allreduce for
    the library's all-reduce).  With RCCL the collective is asynchronous: it is enqueued behind the
    chunk's kernels and overlaps the next chunk's, and nothing is read back before ``finish()``.
    Without a process group (one GPU) it is the identity.  The class knows nothing about kernels:
    ``add`` takes the rank's partial sums as tensors (device tensors from K1 in the product; the
    gloo tests feed it host tensors).
    """

    def __init__(self, nrows, group=None, force=False):
        """``force``: run the collective even in a world of ONE rank (a sum over one rank is the
        identity) -- how the RCCL leg (communicator, asynchronous collective on its own stream,
        ``work.wait()`` ordering) is exercised on a single-GPU box (tests/test_gpu_config4.py)."""
        self.nrows = nrows
        self.group = group
        self._pending = []  # (vector or _Exchange, nt_chunk, has_tail)
        self.active = _in_a_world(group, force)

    def add(self, rows, tail=None):
        """rows: (nrows, nt_chunk) partial sums of this rank; tail: (volo, masso0, area_sum)
        partials, given with the first chunk."""
        rows = rows.to(torch.float64).reshape(self.nrows, -1)
        parts = [rows.reshape(-1)]
        if tail is not None:
            parts += [torch.as_tensor(v, dtype=torch.float64, device=rows.device).reshape(1)
                      for v in tail]
        vec = torch.cat(parts)
        self._pending.append((_Exchange(vec, self.group) if self.active else vec,
                              rows.shape[1], tail is not None))

    def finish(self):
        """-> (rows (nrows, nt) tensor, volo, masso0, area_sum) of the whole grid: float64 host
        tensors after an exchange, the tensors that were added (untouched) without one."""
        out, tail = [], None
        for vec, ntc, has_tail in self._pending:
            if isinstance(vec, _Exchange):
                vec = torch.from_numpy(vec.result())
            n = self.nrows * ntc
            out.append(vec[:n].reshape(self.nrows, ntc))
            if has_tail:
                tail = (vec[n], vec[n + 1], vec[n + 2])
        self._pending = []
        if tail is None:
            raise RuntimeError("ChunkedExchange.finish(): the first chunk's tail was never added")
        return (torch.cat(out, dim=1),) + tail


def _chunk_source(source, device, steps):
    """Normalise the record to an iterator of (t0, t1, T_chunk, S_chunk) device tensors.

    ``source`` is either a pair ``(T, S)`` of (nt,nz,ny_t,nx_t) arrays -- device tensors are
    sliced, host arrays uploaded chunk by chunk (engine.TimeChunks) -- or a tuple
    ``(fetch, nt)`` with ``fetch(t0, t1) -> (T_chunk, S_chunk)`` device tensors, for records that
    exist nowhere in full (bench.py generates chunks with core.synth_field(t0=...)).
    """
    first, second = source
    if callable(first):
        fetch, nt = first, int(second)
        steps = nt if steps is None else max(1, min(int(steps), nt))

        def walk():
            for t0 in range(0, nt, steps):
                t1 = min(t0 + steps, nt)
                Tc, Sc = fetch(t0, t1)
                yield t0, t1, Tc, Sc

        return walk(), nt
    chunks = engine.TimeChunks(first, second, device, steps=steps)
    return iter(chunks), chunks.nt


def steric_global_tile_streamed(source, vol0, areacello, pres, variants=("steric",), steps=None,
                                eos="wright", f32_mode="faithful", group=None, validate_area=True,
                                heat=False, skip_dry=None, events=None, force_collective=False):
    """Global steric of a horizontally tiled grid over a record of ANY length; call on every rank
    with its tile.  The record is walked in time chunks (``steps`` per chunk): K1 -- the
    all-variants kernel when more than one row is wanted -- runs on the chunk and its partial sums
    go into one asynchronous all-reduce per chunk (ChunkedExchange), overlapping the next chunk's
    kernels; ``[volo, masso0, sum(areacello)]`` ride in the first.  The reference state is time
    index 0 of the record, as setup_reference_state builds it.

    source: ``(T, S)`` arrays of this rank's tile or ``(fetch, nt)``, see _chunk_source.
    vol0 (nz,ny_t,nx_t): reference volcello; areacello (ny_t,nx_t).  ``heat``: also the
    ocean-heat-content integrand sum(theta*vol0) (extension).  ``events``: list that receives a
    (start, end) torch.cuda.Event pair per chunk around its K1 launch (bench.py).
    ``force_collective``: all-reduce even with one rank (ChunkedExchange ``force``).
    Returns {variant: finalize() dict} (+ "heat": (nt,) numpy, summed over the whole grid).
    """
    from . import core

    variants = tuple(variants)
    for v in variants:
        if v not in _VARIANTS:
            raise ValueError(f"Unknown variant '{v}' passed to `steric`")
    # the GPU that owns the record (or vol0); host-only operands go to the current device
    owners = [x for x in (source[0], source[1], vol0) if engine._is_device(x)]
    dev = engine.device_of(*owners)
    vol0 = engine.to_device(vol0, dev, torch.float64)
    if not engine.time_dependent(pres):
        pres = engine.to_device(pres, dev, torch.float64)
    one_pass = len(variants) > 1 or heat
    names = list(variants) + (["heat"] if heat else [])
    walker, nt = _chunk_source(source, dev, steps)
    exchange = ChunkedExchange(len(names), group=group, force=force_collective)
    # the held fields of the reference state (time level 0 of the record) are read by the
    # thermosteric / halosteric rows only; a chunk of a RESIDENT record is a view of memory that
    # stays, anything else (uploaded or generated chunks, whose buffers are reused) is cloned --
    # for the steric row of a resident record nothing is copied (2 x 0.93 GB per pass at 0.25 deg)
    need_held = one_pass or variants[0] != "steric"
    views_stay = engine._is_device(source[0]) and engine._is_device(source[1])
    T0 = S0 = None
    first = True
    for t0, t1, Tc, Sc in walker:
        if first and need_held:
            T0, S0 = (Tc[0], Sc[0]) if views_stay else (Tc[0].clone(), Sc[0].clone())
        pc = engine.pressure_chunk(pres, t0, t1, dev)
        ev = None
        if events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            events.append(ev)
        if one_pass:
            rows = core.steric_global_decomp(Tc, Sc, T0, S0, vol0, pc, eos=eos, f32_mode=f32_mode,
                                             skip_dry=skip_dry, events=ev)
            rows = torch.stack([rows[core.DECOMP_ROWS.index(v)] for v in names])
        else:
            Tv, Sv = engine._variant_operands(variants[0], Tc, Sc, T0, S0)
            rows = core.steric_global_masso(Tv, Sv, vol0, pc, eos=eos, f32_mode=f32_mode,
                                            skip_dry=skip_dry, events=ev).reshape(1, -1)
        tail = None
        if first:
            first = False
            # masso0 = masso(t=0) of this very launch: every variant sees (theta0, S0) there, the
            # same kernel, tiling and operands -> steric[t=0] == 0 exactly, for any world size
            area = core.nansum(engine.to_device(areacello, dev, torch.float64))
            tail = (core.nansum(vol0), rows[0, 0], area)
        exchange.add(rows, tail)
    rows, volo, masso0, area = exchange.finish()
    out = {}
    for i, v in enumerate(names):
        if v == "heat":
            out[v] = rows[i].detach().cpu().numpy()
        else:
            out[v] = finalize(rows[i], volo, masso0, area)
    if validate_area and variants:  # util.validate_areacello on the GLOBAL sum, not the tile's
        err = (out[variants[0]]["area_sum"] - 3.6111092e14) / 3.6111092e14
        if not abs(err) < 0.02:
            raise ValueError("Errors found in dataset.")
    return out


def steric_global_tile(T, S, vol0, areacello, pres, variant="steric", eos="wright",
                       f32_mode="faithful", group=None, validate_area=True, steps=None):
    """Global steric of a horizontally tiled grid, device-resident record; call on every rank
    with its tile.  ``steps=None``: the whole record in one K1 launch and one all-reduce;
    otherwise time chunks of ``steps`` (steric_global_tile_streamed).

    T, S: (nt,nz,ny_t,nx_t) device tensors of this rank's tile; vol0 (nz,ny_t,nx_t) the
    reference volcello (time index 0); areacello (ny_t,nx_t).  The reference state is
    time index 0 of (T, S), as setup_reference_state builds it.
    """
    if variant not in _VARIANTS:
        raise ValueError(f"Unknown variant '{variant}' passed to `steric`")
    return steric_global_tile_streamed((T, S), vol0, areacello, pres, variants=(variant,),
                                       steps=steps, eos=eos, f32_mode=f32_mode, group=group,
                                       validate_area=validate_area)[variant]


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


# ---------------------------------------------------------------------------------------------
# the public, labelled API on ONE RANK'S TILE: same signatures as momlevel_amd.steric & co.
# ---------------------------------------------------------------------------------------------
def _sum_over_ranks(group=None, force=False):
    """-> callable summing a float64 numpy vector over the ranks in rank order (identity without
    a group, and in a world of one rank unless ``force``: then the vector goes to the device and
    through the backend's collective all the same -- how a single-GPU box executes the RCCL leg of
    the labelled front end, tests/nccl_worker.py)."""

    def exchange(vec):
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        if not _in_a_world(group, force):
            return vec
        t = torch.from_numpy(vec.copy())
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        return _Exchange(t, group).result()

    return exchange


@accepts_xarray
def steric(dset, reference=None, coord_names=None, varname_map=None, rhozero=1035.0,
           patm=101325.0, equation_of_state="Wright", variant="steric", domain="local",
           dtype="float32", strict=True, annual=False, verbose=False, group=None,
           force_collective=False):
    """``momlevel_amd.steric`` for a horizontally TILED dataset: call it on every rank with the
    rank's own ``(yh, xh)`` tile of thetao / so / volcello / areacello (/ deptho).  Arguments and
    the ``(result, reference)`` return value are those of ``steric`` (src/momlevel/steric.py:17-31).

    Collectives per call (all latency-bound rank-ordered sums of a few doubles, RCCL over xGMI):
      * every domain: an error flag, then sum(areacello) over the tiles (the range check of
        util.validate_areacello is about the whole ocean), then two more error flags around the
        reference state -- a rank that finds its tile unusable makes EVERY rank raise instead of
        leaving the others blocked in the next all-reduce (steric.all_ranks_ok);
      * ``domain="local"``: nothing else -- the DATA path has no collective, the result is the
        tile's part of the field;
      * ``domain="global"``: one more flag and the path's one data exchange, the tile sums
        [masso(t) of every requested variant..., volo] (nt+1 doubles per variant); with a
        time-dependent ``patm`` also [volo, masso] of the reference state.  Every rank returns the
        same global ``reference_height`` and height time series, and volo / masso / rhoga of the
        returned reference state are the GLOBAL ones.
    The bandwidth path for long records is ``steric_global_tile_streamed`` (one asynchronous
    all-reduce per time chunk, nothing else).  ``force_collective``: run the collectives even in a
    world of ONE rank (``_sum_over_ranks``).
    """
    from .steric import _steric_many  # (momlevel_amd.steric the attribute is the function)

    results, reference = _steric_many(
        dset, (variant,), reference, coord_names, varname_map,
        rhozero, patm, equation_of_state, domain, dtype, strict, annual, verbose,
        exchange=_sum_over_ranks(group, force_collective))
    return results[variant], reference


@accepts_xarray
def steric_variants(dset, variants=("steric", "thermosteric", "halosteric"), reference=None,
                    coord_names=None, varname_map=None, rhozero=1035.0, patm=101325.0,
                    equation_of_state="Wright", domain="local", dtype="float32", strict=True,
                    annual=False, verbose=False, heat_content=False, cp=None, group=None,
                    force_collective=False):
    """``momlevel_amd.steric_variants`` on one rank's tile (see ``parallel.steric``): all variants
    (and the heat content) from one pass and ONE all-reduce."""
    from .steric import OHC_CP, _steric_many

    results, reference = _steric_many(
        dset, tuple(variants), reference, coord_names, varname_map,
        rhozero, patm, equation_of_state, domain, dtype, strict, annual, verbose,
        heat_cp=(OHC_CP if cp is None else cp) if heat_content else None,
        exchange=_sum_over_ranks(group, force_collective))
    return results, reference


@accepts_xarray
def setup_reference_state(dset, patm=101325.0, eos="Wright", coord_names=None, time_index=0,
                          group=None, force_collective=False):
    """``momlevel_amd.setup_reference_state`` on one rank's tile: thetao / so / volcello / rho are
    the tile's, volo / masso / rhoga the all-reduced global values."""
    from .reference import _setup
    from .steric import globalise_reference

    ref = _setup(dset, patm, eos, coord_names, time_index, defer_masso=False)
    globalise_reference(ref, _sum_over_ranks(group, force_collective))
    return ref
