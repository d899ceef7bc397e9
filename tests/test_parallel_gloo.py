"""CPU, world_size 2 and 8 over gloo: the N>1 exchange step of the global steric path (8 ranks =
BASELINE.json configs[3]'s 2x4 (yh, xh) layout, which a one-GPU box cannot hold with HIP contexts:
at most 6 processes may use its card at once).

The per-rank partial sums come from the oracle on each rank's horizontal tile
(the kernels need a GPU); the code under test is the product's exchange + epilogue:
momlevel_amd.parallel.exchange_global / finalize and synthetic.tile_bounds.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from momlevel_amd import parallel, synthetic
from oracle import momlevel_numpy as o

NT, NZ, NY, NX = 4, 6, 8, 12


def _case():
    g = synthetic.make_grid(NY, NX, NZ)
    kw = dict(seed=synthetic.SEED, mask3d=g["volcello"])
    T = synthetic.field_numpy((NT, NZ, NY, NX), field_id=1, lo=-2.0, scale=34.0, **kw)
    S = synthetic.field_numpy((NT, NZ, NY, NX), field_id=2, lo=30.0, scale=10.0, **kw)
    return g, T, S


def _tile_partials(g, T, S, rank, world):
    y0, y1, x0, x1 = synthetic.tile_bounds(NY, NX, rank, world)
    pres = o.pressure_from_depth(g["z_l"])
    vol = g["volcello"][:, y0:y1, x0:x1]
    rho = o.calc_rho(T[:, :, y0:y1, x0:x1], S[:, :, y0:y1, x0:x1], pres)
    masso = o.calc_masso(rho, vol)
    return masso, np.nansum(vol), masso[0], np.nansum(g["areacello"][y0:y1, x0:x1])


def _worker(rank, world, port, q):
    os.environ["HIP_VISIBLE_DEVICES"] = ""  # CPU ranks: stay off the card wherever the suite runs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g, T, S = _case()
    masso, volo, masso0, area = _tile_partials(g, T, S, rank, world)
    red = parallel.exchange_global(torch.from_numpy(masso), volo, masso0, area)
    out = parallel.finalize(*red)
    q.put((rank, out["eta"], out["reference_height"], out["volo"], out["area_sum"]))
    parallel.host_barrier()
    dist.destroy_process_group()


def _tile_rows(g, T, S, rank, world):
    """oracle partial sums of this rank's tile for steric / thermosteric / halosteric + heat"""
    y0, y1, x0, x1 = synthetic.tile_bounds(NY, NX, rank, world)
    pres = o.pressure_from_depth(g["z_l"])
    vol = g["volcello"][:, y0:y1, x0:x1]
    Tt, St = T[:, :, y0:y1, x0:x1], S[:, :, y0:y1, x0:x1]
    rows = [o.calc_masso(o.calc_rho(a, b, pres) * np.ones_like(Tt), vol)
            for a, b in ((Tt, St), (Tt, St[0]), (Tt[0], St))]
    rows.append(np.nansum(Tt * vol, axis=(1, 2, 3)))
    return np.stack(rows), np.nansum(vol), np.nansum(g["areacello"][y0:y1, x0:x1])


def _chunked_worker(rank, world, port, q):
    """ChunkedExchange: three time chunks (2+1+1 steps), four rows, the tail in the first chunk"""
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    g, T, S = _case()
    rows, volo, area = _tile_rows(g, T, S, rank, world)
    ex = parallel.ChunkedExchange(4)
    assert ex.active
    for t0, t1 in ((0, 2), (2, 3), (3, 4)):
        tail = (volo, rows[0, 0], area) if t0 == 0 else None
        ex.add(torch.from_numpy(np.ascontiguousarray(rows[:, t0:t1])), tail)
    red, volo_g, masso0_g, area_g = ex.finish()
    outs = [parallel.finalize(red[i], volo_g, masso0_g, area_g) for i in range(3)]
    q.put((rank, [o_["eta"] for o_ in outs], [o_["masso"] for o_ in outs], red[3].numpy(),
           outs[0]["volo"], outs[0]["area_sum"]))
    parallel.host_barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])
def test_exchange_matches_single_domain(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g, T, S = _case()
    vol4 = np.broadcast_to(g["volcello"], T.shape).copy()
    ref, refstate = o.steric(T, S, vol4, g["areacello"], g["z_l"], domain="global")
    for rank, eta, href, volo, area in results:
        assert eta[0] == 0.0
        assert np.isclose(volo, refstate["volo"], rtol=1e-13)
        assert np.isclose(area, 3.6111092e14, rtol=1e-13)
        assert np.isclose(href, ref["reference_height"], rtol=1e-13)
        # eta = h_ref * log(ratio): compare as an expansion coefficient, abs tol 1e-12
        assert np.allclose(eta / href, ref["expansion_coeff"], rtol=0, atol=1e-12)
    # every rank holds the same answer bit for bit
    assert sorted(r[0] for r in results) == list(range(world))
    for r in results[1:]:
        assert np.array_equal(results[0][1], r[1])


def test_exchange_is_identity_without_a_process_group():
    masso = torch.arange(5, dtype=torch.float64)
    m2, v, m0, a = parallel.exchange_global(masso, 2.0, 3.0, 4.0)
    assert torch.equal(m2, masso) and (v.item(), m0.item(), a.item()) == (2.0, 3.0, 4.0)


def test_eight_ranks_tile_the_plane_two_by_four():
    """north_star: "tiled 2x4 across 8xMI355X" -- (yh, xh) = 2 x 4, rank = ry * 4 + rx, x fastest;
    at the 0.25-degree grid every tile is 540 x 360 (contiguous x-runs of 360 cells)"""
    tiles = [synthetic.tile_bounds(1080, 1440, r, 8) for r in range(8)]
    assert tiles[0] == (0, 540, 0, 360) and tiles[3] == (0, 540, 1080, 1440)
    assert tiles[4] == (540, 1080, 0, 360) and tiles[7] == (540, 1080, 1080, 1440)
    cover = np.zeros((1080, 1440), dtype=np.int32)
    for y0, y1, x0, x1 in tiles:
        assert (y1 - y0, x1 - x0) == (540, 360)
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()
    with pytest.raises(ValueError):
        synthetic.tile_bounds(1080, 1442, 0, 8)


def test_rank_ordered_sum_adds_rank_by_rank():
    """every element of the packed vector is reduced in the SAME order -- rank 0, 1, ..., N-1 -- so
    equal per-rank partials at two positions (masso0 and masso(t=0)) give equal sums.  A library
    ring all-reduce reduces segment k starting at rank k: with 8 ranks the 8-rank gloo rehearsal
    measured steric[t=0] = 1.7e-15 instead of 0 (parallel.py, module docstring)."""
    r = np.random.default_rng(5)
    col = r.normal(1e18, 1e17, 8)
    g = np.stack([col, r.normal(0, 1, 8), col], axis=1)  # the same partials at positions 0 and 2
    got = parallel.rank_ordered_sum(g)
    acc = 0.0
    for v in col:
        acc = acc + v if acc else v
    assert got[0] == got[2] == acc
    assert got.dtype == np.float64 and got.shape == (3,)
    # a different rank order of the same numbers differs in the last bits (why the order matters)
    others = {float(parallel.rank_ordered_sum(g[np.roll(np.arange(8), k)])[0]) for k in range(8)}
    assert len(others) > 1


def test_exchange_mode_is_validated(monkeypatch):
    assert parallel.exchange_mode() == "ordered"
    monkeypatch.setenv("MOMLEVEL_AMD_EXCHANGE", "allreduce")
    assert parallel.exchange_mode() == "allreduce"
    monkeypatch.setenv("MOMLEVEL_AMD_EXCHANGE", "ring")
    with pytest.raises(ValueError):
        parallel.exchange_mode()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,exchange", [(2, "ordered"), (8, "ordered"), (8, "allreduce")])
def test_chunked_exchange_all_variants(world, exchange, monkeypatch):
    """one exchange per time chunk, several rows per chunk: result == the single
    domain for every variant, ranks bit-identical; eta[0] == 0 EXACTLY with the default
    rank-ordered sum at any world size -- the library all-reduce (MOMLEVEL_AMD_EXCHANGE=allreduce)
    only promises 1 ulp"""
    monkeypatch.setenv("MOMLEVEL_AMD_EXCHANGE", exchange)  # (spawned workers inherit it)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunked_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g, T, S = _case()
    vol4 = np.broadcast_to(g["volcello"], T.shape).copy()
    for i, variant in enumerate(("steric", "thermosteric", "halosteric")):
        ref, refstate = o.steric(T, S, vol4, g["areacello"], g["z_l"], domain="global",
                                 variant=variant)
        for rank, etas, massos, heat, volo, area in results:
            if exchange == "ordered":
                assert etas[i][0] == 0.0
            else:
                assert abs(etas[i][0] / (volo / area)) <= 1e-14
            assert np.allclose(massos[i], ref["masso"], rtol=1e-13, atol=0)
            href = volo / area
            assert np.allclose(etas[i] / href, ref["expansion_coeff"], rtol=0, atol=1e-12)
        for r in results[1:]:
            assert np.array_equal(results[0][1][i], r[1][i])
    heat = o.ocean_heat_content(T, g["volcello"], 1.0, 1.0)
    assert np.allclose(results[0][3], heat, rtol=1e-13, atol=0)
    for r in results[1:]:
        assert np.array_equal(results[0][3], r[3])


def test_chunked_exchange_without_a_process_group():
    ex = parallel.ChunkedExchange(2)
    assert not ex.active
    ex.add(torch.tensor([[1.0, 2.0], [10.0, 20.0]], dtype=torch.float64), (5.0, 1.0, 7.0))
    ex.add(torch.tensor([[3.0], [30.0]], dtype=torch.float64))
    rows, volo, masso0, area = ex.finish()
    assert rows.tolist() == [[1.0, 2.0, 3.0], [10.0, 20.0, 30.0]]
    assert (volo.item(), masso0.item(), area.item()) == (5.0, 1.0, 7.0)
    with pytest.raises(RuntimeError):
        parallel.ChunkedExchange(1).finish()


def _failing_worker(rank, world, port, q, scenario):
    """the labelled tiled front end when something goes wrong on ONE rank (no GPU needed: the
    failures under test happen before, or instead of, the first kernel)"""
    import warnings

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    import momlevel_amd as m

    d = m.test_data.generate_test_data()
    if scenario == "broken_tile" and rank == 1:
        d = d.drop_vars(["so"]) if hasattr(d, "drop_vars") else _without(d, "so")
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            parallel.steric(d, domain="global", strict=False)
        q.put((rank, "ok", ""))
    except Exception as exc:  # noqa: BLE001
        q.put((rank, type(exc).__name__, str(exc)))
    parallel.host_barrier()  # every rank got here: nobody is stuck in a collective
    dist.destroy_process_group()


def _without(d, name):
    from momlevel_amd.labeled import Dataset

    out = Dataset()
    for k in d.variables:
        if k != name:
            out[k] = d[k]
    return out


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scenario", ["broken_tile", "no_device"])
def test_a_failing_rank_makes_every_rank_raise_instead_of_hanging(scenario):
    """a rank that raised between two collectives used to leave its peers blocked in the
    next all-reduce.  Now the failure is agreed upon first (steric.all_ranks_ok): the failing rank
    raises its own exception, the others a RuntimeError -- and all of them reach the barrier."""
    if scenario == "no_device" and torch.cuda.is_available():
        pytest.skip("needs a GPU-less host: the rank-local failure is the missing device")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q, scenario))
             for r in range(world)]
    for p in procs:
        p.start()
    results = dict((r[0], r[1:]) for r in (q.get(timeout=240) for _ in range(world)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if scenario == "broken_tile":
        assert results[1][0] == "ValueError" and "Errors found in dataset" in results[1][1]
        assert results[0][0] == "RuntimeError" and "other rank" in results[0][1]
    else:  # every rank fails the same way (no HIP device) and says so itself
        assert results[0][0] == results[1][0] == "MomlevelHipError"


def _world_of_one_worker(port, q):
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    dist.init_process_group(backend="gloo", rank=0, world_size=1,
                            init_method=f"tcp://127.0.0.1:{port}")
    payload = np.array([0.1, -2.5e17, 3.0, 1.0e-300, 7.25])
    stats = parallel.exchange_stats
    out = {}
    c0 = stats["collectives"]
    out["unforced"] = parallel._sum_over_ranks()(payload)
    out["unforced_collectives"] = stats["collectives"] - c0
    for mode in ("ordered", "allreduce"):
        os.environ["MOMLEVEL_AMD_EXCHANGE"] = mode
        c0 = stats["collectives"]
        out[mode] = parallel._sum_over_ranks(force=True)(payload)
        out[mode + "_collectives"] = stats["collectives"] - c0
        out[mode + "_last"] = dict(stats["last"])
    # the check the GPU worker relies on must be able to FAIL: with _in_a_world answering False the
    # forced call is the identity again and no collective is counted
    real = parallel._in_a_world
    parallel._in_a_world = lambda group=None, force=False: False
    c0 = stats["collectives"]
    parallel._sum_over_ranks(force=True)(payload)
    out["patched_collectives"] = stats["collectives"] - c0
    parallel._in_a_world = real
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_forced_labelled_exchange_runs_a_collective_in_a_world_of_one():
    """in a world of ONE rank ``_sum_over_ranks`` is the identity and runs no
    collective; ``force=True`` sends the vector through the backend all the same (tests/nccl_worker.py
    does this on the RCCL backend), and ``parallel.exchange_stats`` shows which of the two happened."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world_of_one_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=100)
    p.join(30)
    assert p.exitcode == 0
    payload = np.array([0.1, -2.5e17, 3.0, 1.0e-300, 7.25])
    assert out["unforced_collectives"] == 0 and np.array_equal(out["unforced"], payload)
    for mode in ("ordered", "allreduce"):
        assert out[mode + "_collectives"] == 1
        assert np.array_equal(out[mode], payload)
        assert out[mode + "_last"] == {"backend": "gloo", "mode": mode, "world": 1,
                                       "device": "cpu", "doubles": 5}
    assert out["patched_collectives"] == 0


def test_backend_choice_rehearses_over_gloo_when_ranks_outnumber_gpus():
    """`torchrun --nproc-per-node 2 bench.py --gpus 2` on a one-GPU box died in RCCL ("Duplicate GPU
    detected", gpurun_out/r06_bench_torchrun2.err): ranks that share a card now rehearse over gloo by
    themselves, as the self-launcher's ranks already did; a full node keeps RCCL."""
    assert parallel.choose_backend(8, 8) == "nccl" and parallel.choose_backend(8, 2) == "nccl"
    assert parallel.choose_backend(1, 2) == "gloo" and parallel.choose_backend(0, 4) == "gloo"
    assert parallel.choose_backend(1, 2, "nccl") == "nccl" and parallel.choose_backend(8, 8, "gloo") == "gloo"
