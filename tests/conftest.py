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
def _bulk_transfers_through_pinned_memory():
    """On the GPU box the tests' OWN bulk transfers (``tensor.cpu()``, ``tensor.cuda()`` of a MiB or
    more) go through page-locked memory too.  A pageable copy of that size makes the HIP runtime
    pin the malloc'ed source / destination on the fly and cache the pin by address; both aborts
    this project has seen were GPU page faults on such heap addresses, one of
    them inside a test's ``.cpu()``.  The product moves its data through momlevel_amd.hostio; this
    keeps the checker's side of the suite off the same path."""
    if not _have_gpu():
        yield
        return
    import torch

    limit = 256 << 10
    orig_cpu, orig_cuda = torch.Tensor.cpu, torch.Tensor.cuda

    def cpu(self, *args, **kwargs):
        if self.is_cuda and not args and not kwargs and self.numel() * self.element_size() >= limit:
            out = torch.empty(self.shape, dtype=self.dtype, pin_memory=True)
            out.copy_(self)
            return out
        return orig_cpu(self, *args, **kwargs)

    def cuda(self, *args, **kwargs):
        if (not self.is_cuda and not args and not kwargs and not self.is_pinned()
                and self.numel() * self.element_size() >= limit):
            staged = torch.empty(self.shape, dtype=self.dtype, pin_memory=True)
            staged.copy_(self)
            return orig_cuda(staged)
        return orig_cuda(self, *args, **kwargs)

    torch.Tensor.cpu, torch.Tensor.cuda = cpu, cuda
    yield
    torch.Tensor.cpu, torch.Tensor.cuda = orig_cpu, orig_cuda


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


MIX_KINDS = ["".join(k) for k in __import__("itertools").product("dfw", repeat=3) if k != ("w", "w", "w")]
MIX_FUNCS = ["density", "drho_dtemp", "drho_dsal", "alpha", "beta", "lin_density", "lin_alpha", "lin_beta"]


def mixed_operands(vectors, kinds):
    """(T, S, p) of tests/golden/make_golden.py section (8) for a kinds string such as "ffw": d = the
    float64 array, f = the same array as float32, w = a python float."""
    ops = []
    for i, (k, name) in enumerate(zip(kinds, "TSp")):
        a = vectors[f"mix_{name}"]
        ops.append(float(vectors["mix_weak"][i]) if k == "w"
                   else a.astype(np.float32) if k == "f" else a)
    return ops


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
