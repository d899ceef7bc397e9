"""oracle -- CPU restatement of momlevel's steric hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``momlevel_amd/`` (the product) may import this package.  The
only allowed importers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker / the
timed CPU baseline, never as the thing shipped.

Contents: ``momlevel_numpy`` (the numpy restatement, the oracle proper), ``wright_fused.c`` /
``wright_c`` (fused OpenMP masso of one slab: bench.py's informative CPU line), ``host_abi.c`` /
``host_abi`` (HOST build of the whole C ABI of include/momlevel_hip.h -- SURVEY.md 8b -- a second,
independent restatement in C), ``cpu_worker`` (one process of bench.py's P-process CPU line).

Parity status: PINNED.  ``oracle.momlevel_numpy`` is checked (tests/test_oracle_*.py)
against every tight golden the reference's own tests hold for this path
(tests/golden/reference_goldens.json, transcribed from the reference's
tests/test_wright.py, test_steric.py, test_derived.py, test_util.py) and against
vectors produced in the build container by the reference's own
``src/momlevel/eos/wright.py`` loaded standalone (tests/golden/wright_vectors.npz,
made by tests/golden/make_golden.py).
"""
