"""GPU: the tiled, chunk-streamed global steric path under MORE THAN ONE RANK with the real HIP
kernels (BASELINE.json configs[3] in small): two fresh child processes share the test box's one
GPU, each runs K1 on its horizontal tile through parallel.steric_global_tile_streamed and the
product's per-chunk all-reduce (gloo here, RCCL on a multi-GPU node); the result must be the
single-domain oracle's."""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from momlevel_amd import synthetic
from oracle import momlevel_numpy as o

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(tmp_path, world, nt, nz, ny, nx, steps, mode, dtype):
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        out = str(tmp_path / f"rank{rank}_{mode}_{dtype}.npz")
        outs.append(out)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MOMLEVEL_AMD_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), out, str(nt), str(nz),
             str(ny), str(nx), str(steps), mode, dtype],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [dict(np.load(f)) for f in outs]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,dtype,world", [("generator", "f64", 2), ("resident", "f64", 2),
                                              ("host", "f32", 2), ("generator", "f64", 4)])
def test_tiled_streamed_global_steric_under_several_ranks(tmp_path, mode, dtype, world):
    nt, nz, ny, nx, steps = 11, 6, 16, 24, 4  # chunks of 4+4+3 steps, one all-reduce each
    ranks = _run_world(tmp_path, world, nt, nz, ny, nx, steps, mode, dtype)
    g = synthetic.make_grid(ny, nx, nz)
    npdt = np.float32 if dtype == "f32" else np.float64
    kw = dict(seed=synthetic.SEED, mask3d=g["volcello"], dtype=npdt)
    T = synthetic.field_numpy((nt, nz, ny, nx), field_id=1, lo=-2.0, scale=34.0, **kw)
    S = synthetic.field_numpy((nt, nz, ny, nx), field_id=2, lo=30.0, scale=10.0, **kw)
    vol4 = np.broadcast_to(g["volcello"], T.shape).copy()
    for variant in ("steric", "thermosteric", "halosteric"):
        ref, refstate = o.steric(T, S, vol4, g["areacello"], g["z_l"], domain="global",
                                 variant=variant)
        for r in ranks:
            assert r[f"{variant}_eta"][0] == 0.0
            assert np.max(np.abs(r[f"{variant}_masso"] - ref["masso"]) / ref["masso"]) <= 1e-12
            assert abs(r[f"{variant}_volo"] - refstate["volo"]) <= 1e-12 * refstate["volo"]
            assert abs(r[f"{variant}_area_sum"] - 3.6111092e14) <= 1e-12 * 3.6111092e14
            href = float(r[f"{variant}_reference_height"])
            assert np.allclose(r[f"{variant}_eta"] / href, ref["expansion_coeff"], rtol=0, atol=1e-12)
        for r in ranks[1:]:  # every rank holds the same answer bit for bit
            for k in ("masso", "eta", "volo", "masso0"):
                assert np.array_equal(r[f"{variant}_{k}"], ranks[0][f"{variant}_{k}"])
    heat = o.ocean_heat_content(T, g["volcello"], 1.0, 1.0)
    assert np.max(np.abs(ranks[0]["heat"] - heat) / np.abs(heat)) <= 1e-12
    # the one-pass decomposition and a single-variant run agree bit for bit across the exchange
    assert np.array_equal(ranks[0]["thermo_single_masso"], ranks[0]["thermosteric_masso"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype,world", [("f64", 2), ("f32", 4)])
def test_labelled_tiled_api_matches_the_single_domain_call(tmp_path, dtype, world):
    """momlevel_amd.parallel.steric / steric_variants / setup_reference_state: the reference's
    signatures on ONE RANK'S TILE.  Global results (every rank, bit-identical across ranks) equal
    momlevel_amd.steric on the whole dataset to <= 1e-12 on masso and 1e-12 abs on the expansion
    coefficient; local results are the tile of the whole-domain field, bit for bit."""
    import torch  # noqa: F401  (the parent computes the single-domain answer on the same GPU)

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rank_worker_labelled import tile_dataset

    import momlevel_amd

    nt, nz, ny, nx = 6, 5, 16, 24
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        out = str(tmp_path / f"lab{rank}.npz")
        outs.append(out)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MOMLEVEL_AMD_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "rank_worker_labelled.py"), out, str(nt),
             str(nz), str(ny), str(nx), dtype], env=env, stdout=subprocess.PIPE,
            stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    ranks = [dict(np.load(f)) for f in outs]
    whole, _ = tile_dataset(nt, nz, ny, nx, np.float32 if dtype == "f32" else np.float64, 0, 1)
    for variant in ("steric", "thermosteric"):
        res, ref = momlevel_amd.steric(whole, variant=variant, domain="global")
        h = float(res["reference_height"])
        for r in ranks:
            assert r[f"global_{variant}"][0] == 0.0
            assert abs(float(r[f"global_{variant}_masso"]) - float(ref["masso"])) <= 1e-12 * float(ref["masso"])
            assert abs(float(r[f"global_{variant}_volo"]) - float(ref["volo"])) <= 1e-12 * float(ref["volo"])
            assert abs(float(r[f"global_{variant}_href"]) - h) <= 1e-12 * h
            assert np.allclose(r[f"global_{variant}"] / h, res[variant].values / h, rtol=0, atol=1e-12)
            assert np.array_equal(r[f"global_{variant}"], ranks[0][f"global_{variant}"])
            assert bool(r["untiled_call_raises"])  # a tile alone is not a valid ocean
    lres, lref = momlevel_amd.steric(whole, domain="local")
    vres, _ = momlevel_amd.steric_variants(whole, domain="global", heat_content=True)
    for r in ranks:
        y0, y1, x0, x1 = r["bounds"]
        tile = lres["steric"].values[:, y0:y1, x0:x1]
        assert np.array_equal(np.isnan(r["local_eta"]), np.isnan(tile))
        assert np.array_equal(np.nan_to_num(r["local_eta"]), np.nan_to_num(tile))
        assert abs(float(r["local_ref_masso"]) - float(lref["masso"])) <= 1e-12 * float(lref["masso"])
        assert abs(float(r["setup_masso"]) - float(lref["masso"])) <= 1e-12 * float(lref["masso"])
        hv = float(vres["steric"]["reference_height"])
        for v in ("steric", "thermosteric", "halosteric"):
            assert np.allclose(r[f"variants_{v}"] / hv, vres[v][v].values / hv, rtol=0, atol=1e-12)
        ohc = vres["heat"]["ohc"].values
        assert np.max(np.abs(r["variants_ohc"] - ohc) / np.abs(ohc)) <= 1e-12


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange", ["ordered", "allreduce"])
def test_eight_ranks_two_by_four(tmp_path, exchange):
    """BASELINE.json configs[3]'s layout -- (yh, xh) tiled 2 x 4 over 8 ranks -- in small.  The box
    admits at most 6 processes on its one GPU, so 8 HIP contexts cannot coexist here: THIS process
    runs the product's tile path (K1 through parallel.steric_global_tile_streamed, no process group)
    on each of the 8 tiles in turn and hands every tile's partial sums to one of 8 GPU-less rank
    processes, which carry them through the product's per-chunk exchange over gloo and the
    replicated epilogue (tests/rank_worker_exchange.py).  Against the single-domain oracle and the
    one-GPU product on the whole grid; eta[t=0] == 0 EXACTLY for every variant with the (default)
    rank-ordered exchange -- with the library's all-reduce it was 1.7e-15 at 8 ranks, which is why
    the ordered exchange exists; every rank holds the same bits."""
    import torch

    from momlevel_amd import core, parallel

    world, nt, nz, ny, nx, steps = 8, 11, 6, 16, 24, 4
    g = synthetic.make_grid(ny, nx, nz)
    names = ("steric", "thermosteric", "halosteric")
    dev = torch.device("cuda", torch.cuda.current_device())

    def tile_result(rank, n):
        tile = synthetic.tile_bounds(ny, nx, rank, n)
        th, tw = tile[1] - tile[0], tile[3] - tile[2]
        gt = synthetic.make_grid(ny, nx, nz, tile=tile)
        vol0 = torch.from_numpy(gt["volcello"]).to(dev)
        kw = dict(seed=synthetic.SEED, mask3d=vol0, global_hw=(ny, nx), origin=gt["origin"], device=dev)
        shape = (nt, nz, th, tw)
        T = core.synth_field(shape, torch.float64, field_id=synthetic.FIELD_THETAO,
                             lo=synthetic.THETA_LO, scale=synthetic.THETA_SCALE, **kw)
        S = core.synth_field(shape, torch.float64, field_id=synthetic.FIELD_SO,
                             lo=synthetic.SO_LO, scale=synthetic.SO_SCALE, **kw)
        pres = np.asarray(gt["z_l"]) * 1.0e4 + 101325.0
        return (th, tw), parallel.steric_global_tile_streamed(
            (T, S), vol0, gt["areacello"], pres, variants=names, steps=steps, heat=True,
            validate_area=False)

    assert [synthetic.tile_bounds(ny, nx, r, world) for r in (0, 3, 4, 7)] == [
        (0, 8, 0, 6), (0, 8, 18, 24), (8, 16, 0, 6), (8, 16, 18, 24)]  # 2 x 4, x fastest
    for rank in range(world):
        hw, res = tile_result(rank, world)
        assert hw == (8, 6)
        rows = np.stack([res[v]["masso"] for v in names] + [res["heat"]])
        np.savez(str(tmp_path / f"partials{rank}.npz"), rows=rows, names=np.array(names + ("heat",)),
                 volo=res["steric"]["volo"], area=res["steric"]["area_sum"])
    _, whole = tile_result(0, 1)  # the one-GPU product on the whole grid

    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        out = str(tmp_path / f"rank{rank}.npz")
        outs.append(out)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MOMLEVEL_AMD_EXCHANGE=exchange,
                   HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")  # these ranks never see the GPU
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "rank_worker_exchange.py"),
             str(tmp_path / f"partials{rank}.npz"), out, str(steps)],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    ranks = [dict(np.load(f)) for f in outs]
    assert sorted(int(r["rank"]) for r in ranks) == list(range(world))
    assert not any(bool(r["gpu_open"]) for r in ranks)  # the 8 ranks stayed off the card

    kw = dict(seed=synthetic.SEED, mask3d=g["volcello"], dtype=np.float64)
    T = synthetic.field_numpy((nt, nz, ny, nx), field_id=1, lo=-2.0, scale=34.0, **kw)
    S = synthetic.field_numpy((nt, nz, ny, nx), field_id=2, lo=30.0, scale=10.0, **kw)
    vol4 = np.broadcast_to(g["volcello"], T.shape).copy()
    for variant in names:
        ref, refstate = o.steric(T, S, vol4, g["areacello"], g["z_l"], domain="global",
                                 variant=variant)
        for r in ranks:
            if exchange == "ordered":
                assert r[f"{variant}_eta"][0] == 0.0  # exactly, as on one GPU and in the reference
            else:
                assert abs(r[f"{variant}_expansion_coeff"][0]) <= 1e-14
            assert np.max(np.abs(r[f"{variant}_masso"] - ref["masso"]) / ref["masso"]) <= 1e-12
            assert np.max(np.abs(r[f"{variant}_masso"] - whole[variant]["masso"])
                          / whole[variant]["masso"]) <= 1e-12  # 1 GPU vs 8 ranks
            assert abs(r[f"{variant}_volo"] - refstate["volo"]) <= 1e-12 * refstate["volo"]
            assert abs(r[f"{variant}_area_sum"] - 3.6111092e14) <= 1e-12 * 3.6111092e14
            href = float(r[f"{variant}_reference_height"])
            assert np.allclose(r[f"{variant}_eta"] / href, ref["expansion_coeff"], rtol=0, atol=1e-12)
        for r in ranks[1:]:  # every rank holds the same answer bit for bit
            for k in ("masso", "eta", "volo", "masso0"):
                assert np.array_equal(r[f"{variant}_{k}"], ranks[0][f"{variant}_{k}"])
    heat = o.ocean_heat_content(T, g["volcello"], 1.0, 1.0)
    assert np.max(np.abs(ranks[0]["heat"] - heat) / np.abs(heat)) <= 1e-12
