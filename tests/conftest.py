import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Make sure libmomlevel_hip.so exists (hipcc cross-compiles without a GPU)."""
    from momlevel_amd.csrc import build

    build.build()


@pytest.fixture(scope="session")
def goldens():
    with open(os.path.join(GOLDEN, "reference_goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def wright_vectors():
    return dict(np.load(os.path.join(GOLDEN, "wright_vectors.npz")))


@pytest.fixture(scope="session")
def steric_cases():
    return dict(np.load(os.path.join(GOLDEN, "steric_cases.npz")))


def assert_bit_equal(got, ref, what=""):
    """NaN placement identical; every finite value identical to the last bit."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN mask differs"
    m = ~np.isnan(ref)
    bad = got[m] != ref[m]
    assert not bad.any(), (
        f"{what}: {int(bad.sum())} of {int(m.sum())} finite values differ; "
        f"max rel {np.max(np.abs(got[m][bad] - ref[m][bad]) / np.abs(ref[m][bad])):.3e}"
    )


def assert_rel(got, ref, rtol, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN mask differs"
    m = ~np.isnan(ref)
    if m.any():
        err = np.max(np.abs(got[m] - ref[m]) / np.maximum(np.abs(ref[m]), 1e-300))
        assert err <= rtol, f"{what}: max rel err {err:.3e} > {rtol:.1e}"
