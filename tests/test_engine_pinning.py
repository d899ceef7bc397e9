"""CPU: the host-memory hygiene of engine.TimeChunks' asynchronous uploads:
only the page-aligned INTERIOR of a chunk is ever page-locked, no page is registered twice, a
failed unregister raises instead of leaving a stale pin, and nothing stays registered when an
iteration ends.  The runtime is replaced by a recording stand-in: no GPU needed."""

import warnings

import numpy as np
import pytest

from momlevel_amd import _lib, engine

PAGE = engine._PAGE


def test_page_interior_arithmetic():
    rng = np.random.default_rng(7)
    for _ in range(2000):
        ptr = int(rng.integers(1, 1 << 40))
        nbytes = int(rng.integers(1, 1 << 24))
        lo, hi = engine.page_interior(ptr, nbytes)
        assert lo % PAGE == 0 and hi % PAGE == 0
        assert ptr <= lo < ptr + PAGE
        assert ptr + nbytes - PAGE < hi <= ptr + nbytes
        if hi > lo:  # every registered page lies wholly inside the buffer
            assert lo >= ptr and hi <= ptr + nbytes
    # aligned buffers are registered whole; a buffer inside one page not at all
    assert engine.page_interior(8 * PAGE, 4 * PAGE) == (8 * PAGE, 12 * PAGE)
    lo, hi = engine.page_interior(8 * PAGE + 8, PAGE - 16)
    assert hi <= lo


def test_adjacent_chunks_never_share_a_registered_page():
    """time chunks of one array abut at arbitrary byte offsets: their interiors are disjoint and
    the page that holds the boundary belongs to neither (unless the boundary is page aligned)"""
    rng = np.random.default_rng(11)
    for _ in range(500):
        base = int(rng.integers(1, 1 << 40))
        step = int(rng.integers(1, 1 << 22))  # bytes per time step
        steps = [int(rng.integers(1, 9)) for _ in range(6)]
        edges = np.concatenate([[0], np.cumsum(steps)]) * step + base
        ranges = [engine.page_interior(int(a), int(b - a)) for a, b in zip(edges[:-1], edges[1:])]
        live = [(lo, hi) for lo, hi in ranges if hi > lo]
        for (lo0, hi0), (lo1, hi1) in zip(live[:-1], live[1:]):
            assert hi0 <= lo1
        for (lo, hi), a, b in zip(ranges, edges[:-1], edges[1:]):
            if hi > lo:
                assert a <= lo and hi <= b


class FakeRuntime:
    """mlx_host_pin / mlx_host_unpin with programmable failures"""

    def __init__(self):
        self.pinned = {}
        self.fail_pin = False
        self.fail_unpin = False
        self.calls = []

    def mlx_host_pin(self, ptr, nbytes):
        self.calls.append(("pin", ptr, nbytes))
        assert ptr % PAGE == 0 and nbytes % PAGE == 0 and nbytes > 0, "unaligned registration"
        for s, e in self.pinned.items():
            assert not (ptr < e and s < ptr + nbytes), "a page was registered twice"
        if self.fail_pin:
            return 1
        self.pinned[ptr] = ptr + nbytes
        return 0

    def mlx_host_unpin(self, ptr):
        self.calls.append(("unpin", ptr))
        if self.fail_unpin:
            return 719
        assert ptr in self.pinned, "unregistering a range that is not registered"
        del self.pinned[ptr]
        return 0

    def mlx_last_error(self, buf, n):
        return 0


@pytest.fixture
def runtime(monkeypatch):
    fake = FakeRuntime()
    monkeypatch.setattr(_lib, "load", lambda: fake)
    monkeypatch.setattr(_lib, "last_error", lambda: "simulated")
    monkeypatch.setattr(engine, "_LIVE_PINS", {})
    return fake


def test_no_page_is_pinned_twice(runtime):
    assert engine.pin_interior(16 * PAGE, 32 * PAGE)
    assert not engine.pin_interior(31 * PAGE, 40 * PAGE)  # overlaps the live pin: refused up front
    assert not engine.pin_interior(8 * PAGE, 17 * PAGE)
    assert engine.pin_interior(32 * PAGE, 40 * PAGE)      # abuts, does not overlap
    assert not engine.pin_interior(5 * PAGE, 5 * PAGE)    # empty interior
    assert [c[0] for c in runtime.calls] == ["pin", "pin"]
    engine.unpin_interior(16 * PAGE)
    assert engine.pin_interior(20 * PAGE, 24 * PAGE)      # free again after the release
    engine.unpin_interior(20 * PAGE)
    engine.unpin_interior(32 * PAGE)
    assert runtime.pinned == {} and engine._LIVE_PINS == {}


def test_refused_registration_falls_back_with_a_warning(runtime):
    runtime.fail_pin = True
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert not engine.pin_interior(16 * PAGE, 32 * PAGE)
    assert any("copying synchronously" in str(x.message) for x in w)
    assert engine._LIVE_PINS == {}  # nothing to release later


def test_failed_unregister_raises(runtime):
    assert engine.pin_interior(16 * PAGE, 32 * PAGE)
    runtime.fail_unpin = True
    with pytest.raises(_lib.MomlevelHipError, match="hipHostUnregister failed"):
        engine.unpin_interior(16 * PAGE)


class Event:
    def __init__(self, done):
        self.done = done

    def query(self):
        return self.done

    def synchronize(self):
        self.done = True


def test_release_unpins_finished_uploads_and_leaves_nothing_behind(runtime):
    chunks = object.__new__(engine.TimeChunks)  # the bookkeeping only: no device, no streams
    assert engine.pin_interior(16 * PAGE, 32 * PAGE)
    assert engine.pin_interior(64 * PAGE, 80 * PAGE)
    chunks._registered = [(16 * PAGE, Event(True), None), (64 * PAGE, Event(False), None),
                          (None, Event(False), None)]  # the last: a staging-only upload
    chunks._release()
    assert [r[0] for r in chunks._registered] == [64 * PAGE, None]  # still in flight: kept
    assert list(runtime.pinned) == [64 * PAGE]
    chunks._release(wait=True)
    assert chunks._registered == [] and runtime.pinned == {} and engine._LIVE_PINS == {}
    # a failing unregister surfaces from _release too
    assert engine.pin_interior(16 * PAGE, 32 * PAGE)
    chunks._registered = [(16 * PAGE, Event(True), None)]
    runtime.fail_unpin = True
    with pytest.raises(_lib.MomlevelHipError):
        chunks._release()


def test_leading_one_pressure_is_not_time_dependent():
    """a (1,nz,1,1) pressure is a z profile (core._pressure squeezes it); slicing it per
    time chunk would hand chunk 2 an empty operand"""
    assert not engine.time_dependent(np.zeros((1, 5, 1, 1)))
    assert engine.time_dependent(np.zeros((3, 5, 1, 1)))
    assert not engine.time_dependent(np.zeros((5, 4, 3)))
    p = np.arange(5.0).reshape(1, 5, 1, 1)
    assert engine.pressure_chunk(p, 2, 4, None) is p
