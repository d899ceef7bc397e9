"""GPU: libmomlevel_hip.so driven from plain C++ (examples/c_abi_demo.cpp) -- no Python, no torch on
the calling side -- must print the numbers the Python path computes for the same inputs."""

import os
import re
import shutil
import subprocess

import numpy as np
import pytest
import torch

from momlevel_amd import core

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_demo():
    exe = os.path.join(ROOT, "examples", "c_abi_demo")
    src = os.path.join(ROOT, "examples", "c_abi_demo.cpp")
    deps = [src, os.path.join(ROOT, "include", "momlevel_hip.h"),
            os.path.join(ROOT, "momlevel_amd", "libmomlevel_hip.so")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(map(os.path.getmtime, deps)):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.run(
            [hipcc, "--offload-arch=gfx950", "-O2", src, "-I" + os.path.join(ROOT, "include"),
             "-L" + os.path.join(ROOT, "momlevel_amd"), "-lmomlevel_hip",
             "-Wl,-rpath," + os.path.join(ROOT, "momlevel_amd"), "-o", exe],
            check=True, capture_output=True)
    return exe


def test_c_abi_demo_matches_python_path():
    exe = build_demo()
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "momlevel_amd") + ":" +
               os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], check=True, capture_output=True, text=True, env=env, timeout=120).stdout
    masso_c = np.array([float(x) for x in re.findall(r"masso (\S+)", out)])
    exp_c = np.array([float(x) for x in re.findall(r"expansion (\S+)", out)])
    volo_c = float(re.search(r"volo (\S+)", out).group(1))
    assert "bad dtype -> -3" in out

    nt, nz, ny, nx = 6, 10, 32, 48
    n3 = nz * ny * nx
    i = np.arange(n3)
    vol = np.where((i // 7) % 4 == 0, np.nan, 1.0e9 + 1.0e6 * (i % 1000)).reshape(nz, ny, nx)
    pz = (5.0 + 50.0 * np.arange(nz)) * 1.0e4 + 101325.0
    dvol = torch.from_numpy(vol).cuda()
    T = core.synth_field((nt, nz, ny, nx), seed=20251114, field_id=1, lo=-2.0, scale=34.0, mask3d=dvol)
    S = core.synth_field((nt, nz, ny, nx), seed=20251114, field_id=2, lo=30.0, scale=10.0, mask3d=dvol)
    masso = core.steric_global_masso(T, S, dvol, pz, arith="exact").cpu().numpy()  # the demo passes flags 0
    volo = core.nansum(dvol).item()
    assert np.array_equal(masso_c, masso)  # %.17g round-trips doubles exactly
    assert volo_c == volo
    assert exp_c[0] == 0.0
    assert np.allclose(exp_c, np.log((masso[0] / volo) / (masso / volo)), rtol=0, atol=1e-15)


def test_integration_md_level1_stub_runs_as_printed(wright_vectors):
    """The ctypes stub INTEGRATION.md offers a momlevel maintainer must work verbatim (only the
    library path is made absolute)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(\"\"\" wright_hip\.py.*?)```", text, flags=re.S).group(1)
    code = code.replace('ctypes.CDLL("libmomlevel_hip.so")',
                        f'ctypes.CDLL({os.path.join(ROOT, "momlevel_amd", "libmomlevel_hip.so")!r})')
    ns = {}
    exec(compile(code, "wright_hip.py", "exec"), ns)
    v = wright_vectors
    for name in ("density", "drho_dtemp", "drho_dsal", "alpha", "beta"):
        got = ns[name](v["blk_T"], v["blk_S"], v["blk_p"])
        ref = v[f"blk_{name}"]
        assert got.shape == ref.shape and np.array_equal(got, ref), name
    assert ns["density"](18.0, 35.0, 2.0e5) == v["scalar_out"][0]
