"""Generate the committed golden vectors under tests/golden/.

Run ONCE in the build container, where /root/reference exists:

    python -B tests/golden/make_golden.py

(-B: the process is root, a plain import would drop __pycache__ into the
read-only reference tree.)

Two files are written:

* ``wright_vectors.npz`` -- inputs and the outputs of the REFERENCE'S OWN
  ``src/momlevel/eos/wright.py`` (loaded standalone with importlib: the module
  has no imports, the package itself is not importable here because xarray /
  xgcm / cftime are absent).  These pin the oracle's EOS restatement -- and,
  through it, the HIP kernels -- pointwise and bit for bit.
* ``steric_cases.npz`` -- outputs of ``oracle.momlevel_numpy`` on the
  reference's ``generate_test_data()`` datasets, for the quantities the
  reference's own tests do not pin tightly (the ``domain="global"`` variants:
  tests/test_steric.py:80-125 are atol-dominated).  Labelled "oracle-pinned".

The text of no reference source file is stored: only numbers.
"""

import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

REF_WRIGHT = "/root/reference/src/momlevel/eos/wright.py"
REF_LINEAR = "/root/reference/src/momlevel/eos/linear.py"  # imports numpy only


def load_reference_wright():
    spec = importlib.util.spec_from_file_location("ref_wright", REF_WRIGHT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_linear():
    spec = importlib.util.spec_from_file_location("ref_linear", REF_LINEAR)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_wright()
    out = {}

    # (1) the reference's own test inputs: tests/test_wright.py:4-8
    rng = np.random.default_rng(123)
    out["tw_T"] = rng.normal(15.0, 5.0, (5, 5))
    out["tw_S"] = rng.normal(35.0, 1.5, (5, 5))
    out["tw_p"] = rng.normal(2000.0, 500.0, (5, 5))

    # (2) dense random, oceanographic range and beyond, incl. NaN / inf / zeros
    rng = np.random.default_rng(20251114)
    n = 4096
    T = rng.uniform(-2.0, 32.0, n)
    S = rng.uniform(30.0, 40.0, n)
    p = rng.uniform(101325.0, 6.1e7, n)
    T[:8] = [np.nan, 0.0, -0.0, 40.0, -5.0, np.inf, 15.0, 15.0]
    S[:8] = [35.0, 0.0, 35.0, 45.0, 0.0, 35.0, np.nan, 35.0]
    p[:8] = [2e5, 0.0, 101325.0, 1.1e8, 101325.0, 2e5, 2e5, np.nan]
    out["rnd_T"], out["rnd_S"], out["rnd_p"] = T, S, p

    # (3) a (nt,nz,ny,nx) block with a z-profile pressure, the shape calc_rho sees
    out["blk_T"] = rng.uniform(-2.0, 32.0, (3, 7, 6, 10))
    out["blk_S"] = rng.uniform(30.0, 40.0, (3, 7, 6, 10))
    z_l = np.cumsum(2.0 * 1.075 ** np.arange(7)) * 40.0
    out["blk_p"] = (z_l * 1.0e4 + 101325.0)[:, None, None]

    for tag in ("tw", "rnd", "blk"):
        T, S, p = out[f"{tag}_T"], out[f"{tag}_S"], out[f"{tag}_p"]
        with np.errstate(all="ignore"):
            out[f"{tag}_density"] = ref.density(T, S, p)
            out[f"{tag}_drho_dtemp"] = ref.drho_dtemp(T, S, p)
            out[f"{tag}_drho_dsal"] = ref.drho_dsal(T, S, p)
            out[f"{tag}_alpha"] = ref.alpha(T, S, p)
            out[f"{tag}_beta"] = ref.beta(T, S, p)

    # (4) float32 theta/S with float64 pressure: numpy keeps al0,p0,lam in
    # float32 (python-float constants are weak scalars) -- SURVEY 3.4 #7
    T32 = out["blk_T"].astype(np.float32)
    S32 = out["blk_S"].astype(np.float32)
    out["f32_T"], out["f32_S"] = T32, S32
    out["f32_density"] = ref.density(T32, S32, out["blk_p"])
    assert out["f32_density"].dtype == np.float64
    for fn in ("drho_dtemp", "drho_dsal", "alpha", "beta"):
        out[f"f32_{fn}"] = getattr(ref, fn)(T32, S32, out["blk_p"])
        assert out[f"f32_{fn}"].dtype == np.float64

    # (5) float32 with ONE field held at its first time level -- what steric.py:115-125 hands
    # calc_rho for the thermosteric (S held) and halosteric (theta held) variants: numpy
    # broadcasts the (nz,ny,nx) slab against the (nt,nz,ny,nx) field
    out["f32_density_heldS"] = ref.density(T32, S32[0], out["blk_p"])
    out["f32_density_heldT"] = ref.density(T32[0], S32, out["blk_p"])
    out["f64_density_heldS"] = ref.density(out["blk_T"], out["blk_S"][0], out["blk_p"])
    out["f64_density_heldT"] = ref.density(out["blk_T"][0], out["blk_S"], out["blk_p"])
    for k in ("f32_density_heldS", "f32_density_heldT", "f64_density_heldS", "f64_density_heldT"):
        assert out[k].dtype == np.float64 and out[k].shape == T32.shape

    # (6) the linear EOS (src/momlevel/eos/linear.py:26-162) on the same inputs, float64 and float32
    lin = load_reference_linear()
    for tag, (T, S) in {"tw": (out["tw_T"], out["tw_S"]), "blk": (out["blk_T"], out["blk_S"]),
                        "f32": (T32, S32)}.items():
        out[f"lin_{tag}_density"] = lin.density(T, S)
        out[f"lin_{tag}_alpha"] = lin.alpha(T, S, None)
        out[f"lin_{tag}_beta"] = lin.beta(T, S, None)
    assert out["lin_f32_alpha"].dtype == np.float32  # full_like(T) / density(T, S): float32 throughout
    out["lin_scalar_out"] = np.array([lin.density(18.0, 35.0), lin.drho_dtemp(), lin.drho_dsal(),
                                      lin.alpha(18.0, 35.0, None), lin.beta(18.0, 35.0, None)])

    # (7) round 3: operands no ocean produces -- huge, tiny, infinite, cancelling -- next to ordinary
    # ones, float64 and float32, with a pressure profile of 2e5 / 1e300 / 1e-300 Pa: what the
    # REFERENCE returns there (inf, 0, NaN included) pins the fallback of the kernels' guarded
    # reciprocal (eos_device.hpp) to momlevel itself, not only to the oracle
    r = np.random.default_rng(23)
    nt, nz, ny, nx = 2, 3, 8, 64
    for tag, dtype, big, tiny in (("f64", np.float64, 1e150, 1e-300), ("f32", np.float32, 3e38, 1e-44)):
        T = r.uniform(-2, 32, (nt, nz, ny, nx))
        S = r.uniform(30, 40, (nt, nz, ny, nx))
        weird = [big, -big, big * 1e-3, tiny, 0.0, np.inf, -974.2, 740.54, 1e30, -1e30]
        for i, w in enumerate(weird):
            T[i % nt, i % nz, i % ny, 3 + 5 * i] = w
            S[(i + 1) % nt, i % nz, (i + 3) % ny, 7 + 5 * i] = w
            T[0, (i + 1) % nz, (i + 5) % ny, 11 + 5 * i] = w
            S[0, (i + 1) % nz, (i + 5) % ny, 11 + 5 * i] = -w
        T, S = T.astype(dtype), S.astype(dtype)
        out[f"patho_{tag}_T"], out[f"patho_{tag}_S"] = T, S
        with np.errstate(all="ignore"):
            out[f"patho_{tag}_density"] = ref.density(T, S, np.array([2.0e5, 1.0e300, 1.0e-300])[:, None, None])
        assert out[f"patho_{tag}_density"].dtype == np.float64
    out["patho_p"] = np.array([2.0e5, 1.0e300, 1.0e-300])

    # (8) round 3: every dtype combination numpy's promotion distinguishes.  Each of T, S, p is a
    # float64 array (d), a float32 array (f) or a python float (w: a weak scalar that takes the
    # dtype of the arrays it meets -- calc_pdens' pressure, derived.py:477); the REFERENCE's
    # functions on all 26 combinations with at least one array.  Outputs keep the dtype numpy gave
    # them (float32 where no float64 array takes part).  Pins csrc/eos_promote.hpp.
    m = 512
    out["mix_T"], out["mix_S"], out["mix_p"] = out["rnd_T"][:m], out["rnd_S"][:m], out["rnd_p"][:m]
    out["mix_weak"] = np.array([11.25, 34.7, 2.0e7 + 101325.0])
    import itertools

    for kinds in itertools.product("dfw", repeat=3):
        if kinds == ("w", "w", "w"):
            continue
        ops = []
        for i, (k, name) in enumerate(zip(kinds, "TSp")):
            a = out[f"mix_{name}"]
            ops.append(float(out["mix_weak"][i]) if k == "w" else
                       a.astype(np.float32) if k == "f" else a)
        tag = "mix_" + "".join(kinds)
        with np.errstate(all="ignore"):
            for fn in ("density", "drho_dtemp", "drho_dsal", "alpha", "beta"):
                out[f"{tag}_{fn}"] = np.asarray(getattr(ref, fn)(*ops))
            out[f"{tag}_lin_density"] = np.asarray(lin.density(ops[0], ops[1]))
            out[f"{tag}_lin_alpha"] = np.asarray(lin.alpha(ops[0], ops[1], None))
            out[f"{tag}_lin_beta"] = np.asarray(lin.beta(ops[0], ops[1], None))
    assert out["mix_ffw_density"].dtype == np.float32 and out["mix_fff_alpha"].dtype == np.float32
    assert out["mix_fdd_density"].dtype == np.float64 and out["mix_wfd_density"].dtype == np.float64

    # (9) round 6: eos.linear.density with rho_ref GIVEN (eos/linear.py:55-56): the constant term is
    # (1000 - rho_ref), formed first -- a python float, or a numpy scalar when rho_ref is one -- and
    # then meets the arrays; float64 and float32 fields
    out["linref_values"] = np.array([1035.0, 1027.3, 1035.5, 1035.25])
    refs = {"py": 1035.0, "py2": 1027.3, "np64": np.float64(1035.5), "np32": np.float32(1035.25)}
    for tag, (T, S) in {"tw": (out["tw_T"], out["tw_S"]), "f32": (T32, S32)}.items():
        for rk, rv in refs.items():
            out[f"linref_{tag}_{rk}"] = np.asarray(lin.density(T, S, None, rv))
    assert out["linref_f32_py"].dtype == np.float32 and out["linref_f32_np32"].dtype == np.float32
    assert out["linref_f32_np64"].dtype == np.float64 and out["linref_tw_np32"].dtype == np.float64

    # (10) round 6: a float32 DEPTH coordinate.  steric.py:96 / reference.py:53-54 form
    # pres = dset[zcoord] * 1e4 + patm in the coordinate's dtype: with float32 z_l the pressure is
    # float32, and with float32 theta / S numpy then evaluates the WHOLE equation of state in float32
    z32 = z_l.astype(np.float32)
    out["f32z_z"] = z32
    out["f32z_pres"] = z32 * 1.0e4 + 101325.0
    assert out["f32z_pres"].dtype == np.float32
    out["f32z_density"] = ref.density(T32, S32, out["f32z_pres"][:, None, None])
    out["f32z_density_f64fields"] = ref.density(out["blk_T"], out["blk_S"], out["f32z_pres"][:, None, None])
    assert out["f32z_density"].dtype == np.float32 and out["f32z_density_f64fields"].dtype == np.float64

    # scalars of tests/test_wright.py:11-12,30-31,50-51,70-71,120-121
    out["scalar_args"] = np.array([18.0, 35.0, 200000.0])
    out["scalar_out"] = np.array(
        [
            ref.density(18.0, 35.0, 200000.0),
            ref.drho_dtemp(18.0, 35.0, 200000.0),
            ref.drho_dsal(18.0, 35.0, 200000.0),
            ref.alpha(18.0, 35.0, 200000.0),
            ref.beta(18.0, 35.0, 200000.0),
        ]
    )
    np.savez_compressed(os.path.join(HERE, "wright_vectors.npz"), **out)

    # ---- oracle-pinned steric cases on the reference's test dataset ----------
    from oracle import momlevel_numpy as o

    d = o.generate_test_data()
    cases = {}
    for variant in ("steric", "thermosteric", "halosteric"):
        res, refst = o.steric(
            d["thetao"], d["so"], d["volcello"], d["areacello"], d["z_l"],
            variant=variant, domain="global",
        )
        cases[f"global_{variant}"] = res[variant]
        cases[f"global_{variant}_masso"] = res["masso"]
        cases["global_reference_height"] = np.float64(res["reference_height"])
        res, refst = o.steric(
            d["thetao"], d["so"], d["volcello"], d["areacello"], d["z_l"],
            d["z_i"], d["deptho"], variant=variant, domain="local",
        )
        cases[f"local_{variant}"] = res[variant]
        cases[f"local_{variant}_delta_rho"] = res["delta_rho"]
    cases["ref_rho"] = refst["rho"]
    cases["ref_volo"] = np.float64(refst["volo"])
    cases["ref_masso"] = np.float64(refst["masso"])
    cases["ref_rhoga"] = np.float64(refst["rhoga"])
    np.savez_compressed(os.path.join(HERE, "steric_cases.npz"), **cases)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
