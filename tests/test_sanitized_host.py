"""CPU: the host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer
(SURVEY.md section 5).  scripts/sanitize_host.py builds libmomlevel_hip.so with the HOST pass
instrumented (device code untouched: no GPU sanitizer is used anywhere), fuzzes every entry point
with random dims / strides / NULLs / alignments / enums / flags / workspace sizes from a plain C++
program (tests/native/abi_fuzz.cpp), and re-runs tests/test_abi.py against the sanitized library.
GPU-less processes only: the fuzzer passes fake pointers and refuses to start when a device is
visible."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_c_abi_host_code_is_clean_under_asan_and_ubsan():
    import torch

    if torch.cuda.is_available():
        pytest.skip("fake-pointer fuzzing runs in the GPU-less build container only")
    if os.environ.get("MOMLEVEL_AMD_SANITIZED"):
        pytest.skip("already inside the sanitized run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sanitize_host.py"),
                        "--iters", "20000"], capture_output=True, text=True, timeout=850)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "abi_fuzz OK" in r.stdout and "sanitize_host OK" in r.stdout, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail, tail
