""" steric.py - local and global steric sea level on the MI355X

Drop-in for the public entry points of src/momlevel/steric.py (:17-196): same names,
arguments, defaults, error behaviour and ``(result, reference)`` return value.  What differs
is where the numbers come from -- the fused HIP kernels instead of ~30 numpy passes:

* ``domain="global"``: K1 (``mlx_steric_global``: Wright EOS x reference volcello, summed over
  z,y,x per time step) yields masso(t); the remaining ``log`` on nt scalars stays on the host,
  as in the reference (steric.py:136-142).
* ``domain="local"``: K2 (``mlx_steric_local``: EOS, delta_rho and the dz-weighted column sum in
  one pass, dz evaluated as calc_dz's default path) yields ``delta_rho`` and the height field.

Inputs: labelled datasets backed by numpy (streamed to HBM in time chunks) or by device
tensors (processed in place); xarray Datasets when xarray is installed (adapters.py).
"""

import os

import numpy as np

from . import engine
from .adapters import accepts_xarray
from .labeled import DataArray, Dataset
from .reference import (
    _f32_mode,
    _setup,
    canonical_dims,
    pressure_field,
    pressure_operand,
    refuse_float32_pressure,
    set_reference_masso,
)
from .util import (
    AnnualPlan,
    annual_average,
    default_coords,
    eos_func_from_str,
    validate_dataset,
)

__all__ = ["halosteric", "steric", "steric_variants", "thermosteric"]

_VARIANTS = ("steric", "thermosteric", "halosteric")


def _check_variants(variants):
    for v in variants:
        if v not in _VARIANTS:  # steric.py:125
            raise ValueError(f"Unknown variant '{v}' passed to `steric`")


def _check_dz_inputs(levels, interfaces, depth):
    """calc_dz's sign checks (derived.py:284-292); they stay on the host."""
    assert bool(np.all(np.nan_to_num(depth.values, nan=0.0) >= 0)), (
        "Depth values must all be positive-definite"
    )
    assert bool(np.all(levels.values >= 0)), (
        "Vertical coordinate levels must all be positive-definite"
    )
    assert bool(np.all(interfaces.values >= 0)), (
        "Vertical coordinate interfaces must all be positive-definite"
    )


# EXTENSION (not in momlevel): ocean heat content beside the steric decomposition
# (BASELINE.json configs[4]).  OHC(t) = rhozero * cp * sum(theta * volcello_ref) over the ocean,
# relative to 0 degC, with MOM6's Boussinesq constants.
OHC_CP = 3992.0  # J kg-1 K-1


def _global_results(ops, reference, variants, dtype, tcoord, coords_for, deferred,
                    heat=None, exchange=None, area_total=None):
    """steric.py:134-147 -- masso(t) from K1, then the Boussinesq offline approximation.
    ``heat=(rhozero, cp)``: also the ocean-heat-content extension (key "heat").
    ``exchange`` (tiled multi-GPU runs, momlevel_amd.parallel): sums a float64 vector over the
    ranks -- the path's ONE collective: [masso rows of this tile..., volo of this tile]."""
    T, S, T0, S0, vol0, p, eos = ops

    def local_sums():
        rows = engine.global_masso_variants(T, S, T0, S0, vol0, p, variants, eos=eos,
                                            f32_mode=_f32_mode(), with_heat=heat is not None)
        return {v: m.cpu().numpy() for v, m in rows.items()}  # (synchronises: errors surface here)

    masso, err = _attempt(local_sums)
    all_ranks_ok(exchange, err)  # a rank whose kernels failed must not leave the others waiting
    if exchange is not None:
        names = list(masso)
        nt = masso[names[0]].shape[0]
        vec = np.concatenate([masso[v] for v in names] + [[np.float64(reference["volo"].values)]])
        vec = exchange(vec)
        masso = {v: vec[i * nt:(i + 1) * nt] for i, v in enumerate(names)}
        if deferred:  # a self-made reference state: its volo was this tile's
            reference["volo"].data[...] = vec[-1]
    if deferred:  # the self-generated reference is time index 0 of this very record: every
        # variant sees (theta0, S0) there, so any of them carries masso0 (same kernel, same bits)
        set_reference_masso(reference, masso[variants[0]][0])
    out = {}
    # steric.py:138: reference["volo"] / reference["areacello"].sum() in numpy's dtypes -- a float32
    # areacello sums to a float32 (numpy's own nansum of the 2-D field, on the host), a float32
    # volcello gave a float32 volo (reference.py): then the reference height is a float32 too
    area = reference["areacello"].sum().values
    if area_total is not None:  # tiled run: the all-reduced sum, in the dtype numpy would give it
        area = np.asarray(area_total, dtype=area.dtype)
    for v in variants:
        reference_height, sealevel, _expansion_coeff = engine.global_finalize(
            masso[v], reference["volo"].values, np.float64(reference["rhoga"].values), area)
        result = Dataset()
        result["reference_height"] = DataArray(
            np.asarray(reference_height), (), None,
            {"long_name": "Reference column height", "units": "m"},
        )
        result["reference_height"].encoding["dtype"] = dtype
        result[v] = DataArray(sealevel, (tcoord,), coords_for((tcoord,)))
        out[v] = result
    if heat is not None:
        rhozero, cp = heat
        result = Dataset()
        result["ohc"] = DataArray(
            (np.float64(rhozero) * np.float64(cp)) * masso["heat"], (tcoord,),
            coords_for((tcoord,)),
            {"long_name": "Global ocean heat content relative to 0 degC (momlevel_amd extension)",
             "units": "J",
             "comment": f"rhozero={rhozero} kg m-3 * cp={cp} J kg-1 K-1 * sum(thetao*volcello_ref)"},
        )
        out["heat"] = result
    return out


def _local_results(ops, dset, rho0, variants, dtype, rhozero, names, cdims3, coords_for,
                   plan=None, reference_is_step0=False):
    """steric.py:150-166 -- delta_rho and the column integral from K2.  ``rho0``: the reference
    state's density in canonical (z,y,x) order.  ``plan`` (an util.AnnualPlan): the annual means
    are taken on the device, fused behind K2."""
    T, S, T0, S0, vol0, p, eos = ops
    tcoord, zcoord, zbounds = names
    hdims = cdims3[1:]
    cdims4 = (tcoord,) + cdims3
    deptho = dset["deptho"].transpose(*hdims)
    _check_dz_inputs(dset[zcoord], dset[zbounds], deptho)
    # opt-in (not reference behaviour): MOMLEVEL_AMD_DELTA_RHO=0 leaves the 4-D delta_rho field
    # out of the result -- the kernel then skips its 8 B/cell store and nothing 4-D comes back
    want_delta_rho = os.environ.get("MOMLEVEL_AMD_DELTA_RHO", "1") != "0"
    fields = engine.local_steric_variants(
        T, S, T0, S0, rho0, vol0, p, rhozero, variants,
        z_i=dset[zbounds].data, deptho=deptho.data, eos=eos, f32_mode=_f32_mode(),
        want_delta_rho=want_delta_rho, annual_weights=None if plan is None else plan.weights,
        reference_is_step0=reference_is_step0,
    )
    def result_coords(dims):
        coords = coords_for(dims)
        if plan is not None and tcoord in coords:
            coords[tcoord] = plan.time  # the mid-year time axis of util.py:93-105
        return coords

    out = {}
    for v in variants:
        delta_rho, sealevel = fields[v]
        result = Dataset()
        if want_delta_rho:
            result["delta_rho"] = DataArray(
                delta_rho, cdims4, result_coords(cdims4),
                {"long_name": "change in in situ density from reference state",
                 "units": "kg m-3"},
            )
            result["delta_rho"].encoding["dtype"] = dtype
        result[v] = DataArray(sealevel, (tcoord,) + hdims, result_coords((tcoord,) + hdims))
        out[v] = result
    return out


def all_ranks_ok(exchange, error=None):
    """Make a rank-local failure COLLECTIVE (tiled multi-GPU runs): one tiny all-reduce of an error
    flag, after which either every rank goes on or every rank raises -- the failing rank its own
    exception, the others a RuntimeError naming the situation.  Without it a rank that raised
    between two collectives would leave its peers blocked in the next all-reduce: a hang instead
    of an error.  No-op for single-domain calls (``exchange`` None)."""
    if exchange is None:
        if error is not None:
            raise error
        return
    failed = exchange(np.array([0.0 if error is None else 1.0]))[0]
    if failed > 0:
        if error is not None:
            raise error
        raise RuntimeError(
            f"steric(): {int(failed)} other rank(s) of the tiled run failed before the exchange "
            "(their exception is in their own log); no result on this rank")


def _attempt(fn, *args, **kwargs):
    """-> (result, None) or (None, exception): rank-local work whose failure must first be agreed
    upon by all ranks (all_ranks_ok) before anybody raises"""
    try:
        return fn(*args, **kwargs), None
    except Exception as exc:  # noqa: BLE001 -- re-raised by all_ranks_ok on this very rank
        return None, exc


def globalise_reference(reference, exchange):
    """Replace the tile sums volo / masso of a reference state made from ONE RANK'S tile by the
    sums over all ranks (rhoga follows); scalar reference states only."""
    if reference["masso"].dims:
        return  # the time-dependent state of a patm(time) run: rejected by validation anyway
    vec = exchange(np.array([np.float64(reference["volo"].values),
                             np.float64(reference["masso"].values)]))
    reference["volo"].data[...] = vec[0]
    set_reference_masso(reference, vec[1])


def _steric_many(*args, **kwargs):
    """_steric_body with what the reference state left in flight completed before anything is
    returned or raised (reference._setup sends rho0 to the host asynchronously)."""
    twins = {}
    try:
        return _steric_body(twins, *args, **kwargs)
    finally:
        pending = twins.pop("pending", None)
        if pending is not None:
            import sys

            pending.__exit__(*sys.exc_info())  # (finish(); on an error: drain, keep the error)


def _steric_body(twins, dset, variants, reference, coord_names, varname_map, rhozero, patm,
                 equation_of_state, domain, dtype, strict, annual, verbose, heat_cp=None,
                 exchange=None):
    """The body of steric() for one or several variants sharing one reference state and one
    pass of theta/S through the device.  Returns ({variant: result}, reference).
    ``twins``: the caller's dict for the device tensors of a self-made reference state (and its
    pending rho0 download).
    ``exchange``: None, or -- when ``dset`` is ONE RANK'S horizontal tile of a multi-GPU run
    (momlevel_amd.parallel.steric) -- a callable summing a float64 vector over the ranks; the
    global sums (sum of areacello, volo, masso) then go through it, everything else is local."""
    dset = dset.rename(varname_map)
    names = default_coords(coord_names)
    tcoord, zcoord, zbounds = names

    area_total = None
    extra_vars = None if domain == "global" else [zbounds, "deptho"]
    if exchange is not None:
        # Tiled run: everything a rank can find wrong with ITS tile is established before the
        # first collective and agreed upon by all ranks; only then are the tile sums exchanged.
        from .util import _area_sum, local_findings

        def precheck():
            fatal = local_findings(dset, additional_vars=extra_vars)
            if fatal:
                print("\n".join(fatal))
                raise ValueError("Errors found in dataset.")
            return _area_sum(dset["areacello"])

        tile_area, err = _attempt(precheck)
        all_ranks_ok(exchange, err)
        area_total = float(exchange(np.array([tile_area]))[0])
    # (with the global area every rank reaches the same verdict on the range check)
    validate_dataset(dset, strict=strict, additional_vars=extra_vars, area_total=area_total)
    pres = pressure_field(dset, zcoord, patm)  # 1 m of depth ~ 1 dbar = 1e4 Pa, plus patm
    # (a property of the dataset's dtypes, the same on every rank of a tiled run: all raise, or none)
    refuse_float32_pressure(pres, dset["thetao"], dset["so"], zcoord)

    # (not with a time-dependent patm: that reference state is time dependent itself and is
    #  rejected by the validation below, as in momlevel)
    deferred = reference is None and domain == "global" and tcoord not in pres.dims
    if reference is None:
        # domain="global": masso0 is masso(t=0) of the K1 launch below (same kernel, same bits)
        reference, err = _attempt(_setup, dset, patm, equation_of_state, coord_names, 0,
                                  defer_masso=deferred, twins=twins)
        all_ranks_ok(exchange, err)
        if exchange is not None and not deferred:
            globalise_reference(reference, exchange)
        if verbose:
            print("Generating reference state from first timestep")
    else:
        assert isinstance(reference, Dataset), "`reference` must be an xarray Dataset"
        if verbose:
            print("Using supplied reference state")
    _, err = _attempt(validate_dataset, reference, reference=True, strict=strict,
                      area_total=area_total)
    all_ranks_ok(exchange, err)

    _check_variants(variants)
    eos_func_from_str(equation_of_state)  # unknown EOS -> ValueError (util.py:247)

    # canonical (time, z, y, x) layout; outputs are always time-first (steric.py:154,165)
    cdims3 = canonical_dims(dset["thetao"], tcoord, zcoord)
    cdims4 = (tcoord,) + cdims3

    def coords_for(dims):
        return {d: dset[d] for d in dims if d in dset.variables}

    def slab(name):
        """a (z,y,x) field of the reference state in canonical order: the device tensor the state
        was computed from when it was made in this call, else the Dataset's array"""
        return twins[name] if name in twins else reference[name].transpose(*cdims3).data

    ops = (
        dset["thetao"].transpose(*cdims4).data,
        dset["so"].transpose(*cdims4).data,
        slab("thetao"),
        slab("so"),
        slab("volcello"),
        pressure_operand(pres, tcoord, cdims3),
        equation_of_state.lower(),
    )
    plan = None
    if annual and domain != "global":
        plan = AnnualPlan(dset[tcoord], tcoord)  # asserts 12 steps per year (util.py:85)
        if not plan.contiguous:
            plan = None  # unusual time axis: average on the host afterwards
    if domain == "global":
        results = _global_results(ops, reference, variants, dtype, tcoord, coords_for, deferred,
                                  heat=None if heat_cp is None else (rhozero, heat_cp),
                                  exchange=exchange, area_total=area_total)
    else:
        if heat_cp is not None:
            raise ValueError("heat_content is a global integral: use domain='global'")
        # (twins: the reference state was made in this call from time level 0 of this record)
        results = _local_results(ops, dset, slab("rho"), variants, dtype, rhozero, names, cdims3,
                                 coords_for, plan, reference_is_step0="thetao" in twins)

    for variant, result in results.items():
        if variant == "heat":
            if annual:
                results[variant] = annual_average(result)
            continue
        result[variant].attrs.update(
            {"long_name": f"{variant.capitalize()} height adjustment", "units": "m"}
        )
        result[variant].encoding["dtype"] = dtype
        # coordinate / dimension attributes follow the input dataset (steric.py:177-179)
        for var in set(result.coords).union(result.dims):
            if var in dset.variables and var in result.variables:
                result[var].attrs.update(dset[var].attrs)
        if annual and plan is None:
            results[variant] = annual_average(result)
    return results, reference


@accepts_xarray
def steric(
    dset,
    reference=None,
    coord_names=None,
    varname_map=None,
    rhozero=1035.0,
    patm=101325.0,
    equation_of_state="Wright",
    variant="steric",
    domain="local",
    dtype="float32",
    strict=True,
    annual=False,
    verbose=False,
):
    """Steric, thermosteric or halosteric sea level change relative to a reference state.

    Parameters (as the reference, src/momlevel/steric.py:33-82)
    ----------
    dset : Dataset with thetao, so, volcello (t,z,y,x) and areacello (y,x); for
        ``domain="local"`` also the z bounds coordinate and ``deptho``
    reference : Dataset, optional -- a reference state from an earlier call; time index 0 of
        ``dset`` is used when omitted
    coord_names : dict, optional -- ``{"t":…, "z":…, "zbounds":…}`` overrides of
        ``("time", "z_l", "z_i")``
    varname_map : dict, optional -- variables renamed before anything else
    rhozero : float -- Boussinesq reference density, kg m-3 (local variant)
    patm : float or DataArray -- sea-surface atmospheric pressure, Pa
    equation_of_state : str -- a module of ``momlevel_amd.eos`` ("Wright", "linear")
    variant : "steric" | "thermosteric" | "halosteric"
    domain : "local" (column integral per grid point) | "global" (one value per time step)
    dtype : str -- output ENCODING metadata only; the arithmetic is float64
    strict : bool -- False downgrades the areacello range check to a warning
    annual : bool -- days-in-month weighted annual means of the result
    verbose : bool

    Returns
    -------
    (result, reference) : tuple of Datasets
    """
    results, reference = _steric_many(
        dset, (variant,), reference, coord_names, varname_map, rhozero, patm, equation_of_state,
        domain, dtype, strict, annual, verbose,
    )
    return (results[variant], reference)


@accepts_xarray
def steric_variants(
    dset,
    variants=("steric", "thermosteric", "halosteric"),
    reference=None,
    coord_names=None,
    varname_map=None,
    rhozero=1035.0,
    patm=101325.0,
    equation_of_state="Wright",
    domain="local",
    dtype="float32",
    strict=True,
    annual=False,
    verbose=False,
    heat_content=False,
    cp=OHC_CP,
):
    """EXTENSION (not in momlevel): several variants in one call.

    ``steric``, ``thermosteric`` and ``halosteric`` of the same dataset are usually wanted
    together, and with host-resident inputs each call is bound by moving theta/S over PCIe.
    This entry point uploads every time chunk once and runs the requested variants on it, with
    one shared reference state.  Each result is bit-identical to the corresponding single call.
    With ``domain="global"`` two or more variants come out of ONE pass of the all-variants kernel
    (theta/S read once).  ``heat_content=True`` (global only) adds ``results["heat"]["ohc"]``, the
    ocean heat content ``rhozero * cp * sum(thetao * volcello_ref)`` per time step from the same
    pass -- an extension with no counterpart in momlevel.

    Returns
    -------
    (results, reference) : ``results`` maps variant name -> result Dataset
    """
    results, reference = _steric_many(
        dset, tuple(variants), reference, coord_names, varname_map, rhozero, patm,
        equation_of_state, domain, dtype, strict, annual, verbose,
        heat_cp=cp if heat_content else None,
    )
    return (results, reference)


def halosteric(*args, **kwargs):
    """``steric(..., variant="halosteric")``: salinity varies, temperature held at the reference."""
    return steric(*args, **kwargs, variant="halosteric")


def thermosteric(*args, **kwargs):
    """``steric(..., variant="thermosteric")``: temperature varies, salinity held."""
    return steric(*args, **kwargs, variant="thermosteric")
