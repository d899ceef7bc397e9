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
