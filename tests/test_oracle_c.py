"""CPU: the C restatement (oracle/wright_fused.c) against the numpy oracle and the reference
module's own vectors -- two independent restatements must agree bit for bit."""

import numpy as np

from oracle import momlevel_numpy as o
from oracle import wright_c as c
from conftest import assert_bit_equal


def test_c_density_matches_reference_vectors(wright_vectors):
    v = wright_vectors
    T, S = v["blk_T"], v["blk_S"]
    pz = v["blk_p"].reshape(-1)
    for t in range(T.shape[0]):
        assert_bit_equal(c.density_slab(T[t], S[t], pz), v["blk_density"][t], f"slab {t}")


def test_c_fused_masso_matches_numpy_oracle():
    r = np.random.default_rng(11)
    shape = (9, 33, 47)
    T = r.uniform(-2, 32, shape)
    S = r.uniform(30, 40, shape)
    vol = r.uniform(1e9, 1e11, shape)
    land = r.uniform(size=shape) < 0.3
    vol[land] = np.nan
    T[land] = np.nan
    pz = o.pressure_from_depth(np.linspace(1.0, 5000.0, shape[0]))
    ref = o.calc_masso(o.calc_rho(T, S, pz), vol)
    got = c.masso_slab(T, S, vol, pz)
    assert abs(got - ref) / ref < 1e-13
    assert c.masso_slab(T, S, np.full(shape, np.nan), pz) == 0.0
    assert c.num_threads() >= 1
