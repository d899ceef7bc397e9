"""numpy restatement of momlevel's steric hot path.  TEST INFRASTRUCTURE ONLY.

This is the CPU oracle of the MI355X build: an op-for-op numpy restatement of
the reference algorithm, written from the reference's formulas and semantics
(file:line citations are into /root/reference, which is NOT needed at run
time).  It deliberately keeps the reference's *unfused* evaluation -- one numpy
pass and one temporary per arithmetic operator -- so that

* its results are bit-identical to what numpy computes in the reference, and
* timing it (bench.py ``cpu_baseline``, kind "port") reproduces the memory
  behaviour of the reference's CPU path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (momlevel_amd/) never does.

xarray semantics that are restated by hand here (xarray is not installed):
broadcast-by-name, ``skipna=True`` reductions (NaN counts as 0, an all-NaN
reduction gives 0.0), ``xr.where`` and ``transpose(time, ...)``.
All arrays are C-contiguous ``(time, z_l, yh, xh)`` / ``(z_l, yh, xh)`` / ``(yh, xh)``.
"""

import numpy as np

# ----------------------------------------------------------------------------
# Wright (1997) equation of state -- src/momlevel/eos/wright.py:6-20 (constants)
# ----------------------------------------------------------------------------
A0 = 7.057924e-4
A1 = 3.480336e-7
A2 = -1.112733e-7
B0 = 5.790749e8
B1 = 3.516535e6
B2 = -4.002714e4
B3 = 2.084372e2
B4 = 5.944068e5
B5 = -9.643486e3
C0 = 1.704853e5
C1 = 7.904722e2
C2 = -7.984422
C3 = 5.140652e-2
C4 = -2.302158e2
C5 = -3.079464


def _wright_terms(T, S):
    """al0, p0, lam of src/momlevel/eos/wright.py:44-46 (same operator order)."""
    al0 = A0 + A1 * T + A2 * S
    p0 = B0 + B4 * S + T * (B1 + T * (B2 + B3 * T) + B5 * S)
    lam = C0 + C4 * S + T * (C1 + T * (C2 + C3 * T) + C5 * S)
    return al0, p0, lam


def wright_density(T, S, p):
    """In-situ density, src/momlevel/eos/wright.py:23-50."""
    al0, p0, lam = _wright_terms(T, S)
    I_denom = 1.0 / (lam + al0 * (p + p0))
    return (p + p0) * I_denom


def wright_drho_dtemp(T, S, p):
    """d(rho)/d(theta), src/momlevel/eos/wright.py:53-85."""
    al0, p0, lam = _wright_terms(T, S)
    I_denom2 = 1.0 / (lam + al0 * (p + p0))
    I_denom2 = I_denom2 * I_denom2
    return I_denom2 * (
        lam * (B1 + T * (2.0 * B2 + 3.0 * B3 * T) + B5 * S)
        - (p + p0) * ((p + p0) * A1 + (C1 + T * (C2 * 2.0 + C3 * 3.0 * T) + C5 * S))
    )


def wright_drho_dsal(T, S, p):
    """d(rho)/d(S), src/momlevel/eos/wright.py:88-119."""
    al0, p0, lam = _wright_terms(T, S)
    I_denom2 = 1.0 / (lam + al0 * (p + p0))
    I_denom2 = I_denom2 * I_denom2
    return I_denom2 * (
        lam * (B4 + B5 * T) - (p + p0) * ((p + p0) * A2 + (C4 + C5 * T))
    )


def wright_alpha(T, S, p):
    """Thermal expansion coefficient, src/momlevel/eos/wright.py:122-142."""
    return -1.0 * (wright_drho_dtemp(T, S, p) / wright_density(T, S, p))


def wright_beta(T, S, p):
    """Haline contraction coefficient, src/momlevel/eos/wright.py:145-165."""
    return wright_drho_dsal(T, S, p) / wright_density(T, S, p)


# linear EOS -- src/momlevel/eos/linear.py:17-23,55-56
LIN_RHO_T0_S0 = 1000.0
LIN_DRHO_DT = -0.2
LIN_DRHO_DS = 0.8


def linear_density(T, S, p=None, rho_ref=None):
    """Linear EOS density, src/momlevel/eos/linear.py:26-58."""
    rho = LIN_RHO_T0_S0 if rho_ref is None else (LIN_RHO_T0_S0 - rho_ref)  # :55
    return rho + ((LIN_DRHO_DT * T) + (LIN_DRHO_DS * S))  # :56


def linear_drho_dtemp(T=None, S=None, p=None):
    """src/momlevel/eos/linear.py:61-84: the constant DRHO_DT."""
    return LIN_DRHO_DT


def linear_drho_dsal(T=None, S=None, p=None):
    """src/momlevel/eos/linear.py:87-110: the constant DRHO_DS."""
    return LIN_DRHO_DS


def linear_alpha(T, S, p=None):
    """src/momlevel/eos/linear.py:113-136."""
    return -1.0 * (np.full_like(T, fill_value=LIN_DRHO_DT) / linear_density(T, S, p))


def linear_beta(T, S, p=None):
    """src/momlevel/eos/linear.py:139-162."""
    return np.full_like(T, fill_value=LIN_DRHO_DS) / linear_density(T, S, p)


_EOS = {
    "wright": {
        "density": wright_density,
        "drho_dtemp": wright_drho_dtemp,
        "drho_dsal": wright_drho_dsal,
        "alpha": wright_alpha,
        "beta": wright_beta,
    },
    "linear": {
        "density": linear_density,
        "drho_dtemp": linear_drho_dtemp,
        "drho_dsal": linear_drho_dsal,
        "alpha": linear_alpha,
        "beta": linear_beta,
    },
}


def eos_func_from_str(eos_str, func_name="density"):
    """src/momlevel/util.py:227-249 -- lower-case the name, unknown => ValueError."""
    assert isinstance(eos_str, str), "Expecting string for equation of state"
    eos_str = eos_str.lower()
    if eos_str not in _EOS:
        raise ValueError(f"Unknown equation of state: {eos_str}")
    return _EOS[eos_str][func_name]


# ----------------------------------------------------------------------------
# derived.py hot subset
# ----------------------------------------------------------------------------
def nansum(x, axis=None):
    """xarray's default ``.sum()``: skipna=True, min_count=None => ``np.sum(where(isnull(x), 0, x))``
    IN THE ARRAY'S OWN DTYPE, which is what numpy.nansum does -- so a float32 volcello / areacello
    (what MOM6 writes) sums to a float32, accumulated in float32 (derived.py:789, steric.py:138)."""
    return np.nansum(x, axis=axis)


def pressure_from_depth(z_l, patm=101325.0):
    """src/momlevel/steric.py:93-96 / reference.py:53-54: 1e4 Pa per metre + patm -- IN THE
    COORDINATE'S DTYPE, as ``dset[zcoord] * 1e4 + patm`` is: a float32 z_l gives a float32 pressure
    (and float32 theta / S then a float32 density throughout)."""
    return (np.asarray(z_l) * 1.0e4) + patm


def calc_rho(thetao, so, pres, eos="Wright"):
    """src/momlevel/derived.py:597-639.

    ``pres`` is the z-profile ``(nz,)`` (apply_ufunc hands the kernel ``(nz,1,1)``),
    a scalar, or an array already broadcastable against thetao/so.
    """
    func = eos_func_from_str(eos)
    pres = np.asarray(pres)
    if pres.ndim == 1:
        pres = pres[:, None, None]
    return func(thetao, so, pres)


def calc_pdens(thetao, so, level=0.0, patm=101325, eos="Wright"):
    """src/momlevel/derived.py:447-486."""
    assert 0.0 <= level <= 7500.0, "specified level must be between 0 and 7500 m"
    return calc_rho(thetao, so, (level * 1.0e4) + patm, eos=eos)


def inverse_barometer(tos, sos, pso, gravity=9.8, equation_of_state="Wright"):
    """src/momlevel/dynamic.py:8-41."""
    rho_conv = calc_rho(tos, sos, pso, eos=equation_of_state)
    return pso * (-1.0 / (rho_conv * gravity))


# ----------------------------------------------------------------------------
# Stratification diagnostics: the consumers of alpha / beta
# ----------------------------------------------------------------------------
def differentiate_z(f, z, axis):
    """``DataArray.differentiate(zcoord, edge_order=2)`` (src/momlevel/derived.py:399-400,
    :752-753): xarray hands the data and the coordinate values to ``numpy.gradient`` -- the
    third-party routine itself, called here as the reference calls it (second-order one-sided
    differences at the two ends; a float32 field gives a float32 derivative)."""
    return np.gradient(f, np.asarray(z), axis=axis, edge_order=2)


def calc_n2(thetao, so, z_l, eos="Wright", gravity=-9.8, patm=101325.0, zaxis=-3):
    """src/momlevel/derived.py:328-411 with ``interfaces=None`` (cell centres; the
    ``interfaces`` branch needs xgcm).  thetao / so: (..., z, y, x); z_l: (nz,)."""
    z_l = np.asarray(z_l, dtype=np.float64)
    pres = ((z_l * 1.0e4) + patm)[:, None, None]  # derived.py:396: thetao[zcoord] * 1e4 + patm
    alpha = eos_func_from_str(eos, func_name="alpha")(thetao, so, pres)
    beta = eos_func_from_str(eos, func_name="beta")(thetao, so, pres)
    dtdz = differentiate_z(thetao, z_l, zaxis)
    dsdz = differentiate_z(so, z_l, zaxis)
    return gravity * ((alpha * dtdz) - (beta * dsdz))  # derived.py:401


def adjust_negative_n2(n2):
    """src/momlevel/derived.py:30-71, quirk included: ``adjusted[0]`` indexes the LEADING
    dimension of the array, whatever it is -- the time axis for a (time, z, y, x) field, the
    surface only for a (z, y, x) one.  The forward fill runs along z (axis -3)."""
    n2 = np.asarray(n2)
    mask = np.where(np.isnan(n2), np.nan, 1.0)
    with np.errstate(invalid="ignore"):
        adjusted = np.where(n2 <= 0.0, np.nan, n2)
    adjusted[0] = np.where(np.isnan(adjusted[0]), 1.0e-8, adjusted[0])
    zaxis = adjusted.ndim - 3
    a = np.moveaxis(adjusted, zaxis, 0).copy()
    for k in range(1, a.shape[0]):  # ffill(zcoord)
        a[k] = np.where(np.isnan(a[k]), a[k - 1], a[k])
    adjusted = np.moveaxis(a, 0, zaxis)
    return adjusted * mask


def calc_stability_angle(thetao, so, pres, z_l, eos="Wright", zaxis=-3):
    """src/momlevel/derived.py:714-766 (the Turner angle, degrees).  ``pres`` broadcasts against
    the fields (the reference's test passes the z profile z_l * 1e4)."""
    pres = np.asarray(pres)
    if pres.ndim == 1:
        pres = pres[:, None, None]
    alpha = eos_func_from_str(eos, func_name="alpha")(thetao, so, pres)
    beta = eos_func_from_str(eos, func_name="beta")(thetao, so, pres)
    dtdz = differentiate_z(thetao, z_l, zaxis)
    dsdz = differentiate_z(so, z_l, zaxis)
    with np.errstate(divide="ignore", invalid="ignore"):
        R_rho = (beta * dsdz) / (alpha * dtdz)
        return np.degrees(np.arctan((1 + R_rho) / (1 - R_rho)))


def calc_wave_speed(n2, dz):
    """src/momlevel/derived.py:798-831 for a (z, y, x) n2 (one time level: there ``n2[0]`` IS the
    surface, as the function's text intends; with a leading time axis the reference's
    ``xr.where(n2[0].isnull(), ...)`` broadcasts (z,y,x) against (time,y,x) into a 4-D array --
    the golden tests/test_derived.py:147-151 is the sum of that array and is reproduced by
    calc_wave_speed_4d_quirk below)."""
    with np.errstate(invalid="ignore"):
        result = nansum(np.sqrt(adjust_negative_n2(n2)) * dz, axis=-3) / np.pi
    return np.where(np.isnan(n2[0]), np.nan, result)


def calc_wave_speed_4d_quirk(n2, dz):
    """The reference's result for a (time, z, y, x) n2, as xarray broadcasts it: dims
    (z, y, x, time) -- condition n2[time=0] (z,y,x) against the (time,y,x) speeds."""
    with np.errstate(invalid="ignore"):
        result = nansum(np.sqrt(adjust_negative_n2(n2)) * dz, axis=-3) / np.pi  # (time, y, x)
    cond = np.isnan(n2[0])  # (z, y, x)
    return np.where(cond[..., None], np.nan, np.moveaxis(result, 0, -1)[None])


def calc_volo(volcello):
    """src/momlevel/derived.py:769-795 -- asserts 3-D, skipna sum."""
    assert volcello.ndim == 3, "Expecting only 3 dimensions for volcello"
    return nansum(volcello)


def calc_masso(rho, volcello):
    """src/momlevel/derived.py:414-444 -- sum(rho*volcello) over every non-time dim.

    rho is (nt,nz,ny,nx) or (nz,ny,nx); volcello is (nz,ny,nx) (broadcast over
    time) or the same shape as rho.
    """
    masso = rho * volcello
    if masso.ndim == 4:
        return nansum(masso, axis=(1, 2, 3))
    return nansum(masso)


def calc_rhoga(masso, volo):
    """src/momlevel/derived.py:642-666."""
    return masso / volo


def calc_dz(levels, interfaces, depth, top=0.0, bottom=None, fraction=False):
    """Partial-bottom-cell thickness, src/momlevel/derived.py:249-325.

    Returns ``(nz, ny, nx)`` (the reference's broadcast order is (yh, xh, z_l);
    only the element values matter downstream).
    """
    levels = np.asarray(levels, dtype=np.float64)
    interfaces = np.asarray(interfaces, dtype=np.float64)
    depth = np.asarray(depth, dtype=np.float64)
    assert bool(np.all(np.nan_to_num(depth, nan=0.0) >= 0)), (
        "Depth values must all be positive-definite"
    )
    assert bool(np.all(levels >= 0)), (
        "Vertical coordinate levels must all be positive-definite"
    )
    assert bool(np.all(interfaces >= 0)), (
        "Vertical coordinate interfaces must all be positive-definite"
    )
    depth = np.where(np.isnan(depth), 0.0, depth)
    if bottom is not None:
        depth = np.minimum(depth, bottom)
    ztop = interfaces[0:-1][:, None, None]
    zbot = interfaces[1:][:, None, None]
    depth = depth[None, :, :]
    dz_field = zbot - ztop
    part = depth - ztop
    part = np.where(part < 0.0, 0.0, part)
    result = np.minimum(part, dz_field)
    part = zbot - top
    part = np.where(part < 0.0, 0.0, part)
    result = np.minimum(part, result)
    if fraction:
        _dz_field = np.where(dz_field == 0, np.nan, dz_field)
        _dz_part = np.where(result == 0, np.nan, result)
        result = _dz_part / _dz_field
    return np.ascontiguousarray(np.broadcast_to(result, (len(levels),) + depth.shape[1:]))


# ----------------------------------------------------------------------------
# util.py hot subset
# ----------------------------------------------------------------------------
def default_coords(coord_names=None):
    """src/momlevel/util.py:199-224."""
    coord_names = {} if coord_names is None else coord_names
    assert isinstance(coord_names, dict), "Coordinate mapping must be a dictionary."
    zcoord = coord_names["z"] if "z" in coord_names.keys() else "z_l"
    zbounds = coord_names["zbounds"] if "zbounds" in coord_names.keys() else "z_i"
    tcoord = coord_names["t"] if "t" in coord_names.keys() else "time"
    return (tcoord, zcoord, zbounds)


def validate_areacello(areacello, reference=3.6111092e14, tolerance=0.02):
    """src/momlevel/util.py:669-694."""
    error = (nansum(areacello) - reference) / reference
    return bool(np.abs(error) < tolerance)


# ----------------------------------------------------------------------------
# reference.py / steric.py on plain arrays
# ----------------------------------------------------------------------------
def setup_reference_state(
    thetao, so, volcello, areacello, z_l, patm=101325.0, eos="Wright", time_index=0
):
    """src/momlevel/reference.py:15-85 on plain arrays; returns a dict."""
    pres = pressure_from_depth(z_l, patm)
    ref = {}
    ref["thetao"] = np.array(thetao[time_index])
    ref["so"] = np.array(so[time_index])
    ref["volcello"] = np.array(volcello[time_index])
    ref["rho"] = calc_rho(ref["thetao"], ref["so"], pres, eos=eos)
    ref["volo"] = calc_volo(ref["volcello"])
    ref["masso"] = calc_masso(ref["rho"], ref["volcello"])
    ref["rhoga"] = calc_rhoga(ref["masso"], ref["volo"])
    ref["areacello"] = np.asarray(areacello)
    return ref


def steric(
    thetao,
    so,
    volcello,
    areacello,
    z_l,
    z_i=None,
    deptho=None,
    reference=None,
    rhozero=1035.0,
    patm=101325.0,
    equation_of_state="Wright",
    variant="steric",
    domain="local",
    strict=True,
):
    """src/momlevel/steric.py:17-184 on plain arrays.

    Returns ``(result, reference)`` dicts.  global: result has
    ``reference_height`` (scalar), ``<variant>`` (nt,) and -- extra, for tests --
    ``masso`` (nt,) and ``expansion_coeff`` (nt,).  local: ``delta_rho``
    (nt,nz,ny,nx) and ``<variant>`` (nt,ny,nx).
    """
    if not validate_areacello(areacello):  # util.py:783-792
        msg = "Variable `areacello` field is out of range. It may not be masked."
        if strict:
            raise ValueError("Errors found in dataset.")
        import warnings

        warnings.warn(msg)

    pres = pressure_from_depth(z_l, patm)  # steric.py:96

    if reference is None:  # steric.py:98-109
        reference = setup_reference_state(
            thetao, so, volcello, areacello, z_l, patm=patm, eos=equation_of_state
        )

    # steric.py:115-125 -- which field is held at the reference state
    if variant == "thermosteric":
        T, S = thetao, reference["so"]
    elif variant == "halosteric":
        T, S = reference["thetao"], so
    elif variant == "steric":
        T, S = thetao, so
    else:
        raise ValueError(f"Unknown variant '{variant}' passed to `steric`")

    nt = thetao.shape[0]
    rho = calc_rho(T, S, pres, eos=equation_of_state)  # steric.py:128
    if rho.ndim == 3:  # both fields held (cannot happen via the public API)
        rho = np.broadcast_to(rho, (nt,) + rho.shape)

    result = {}
    if domain == "global":  # steric.py:134-147
        masso = calc_masso(rho, reference["volcello"])
        expansion_coeff = np.log(reference["rhoga"] / (masso / reference["volo"]))
        reference_height = reference["volo"] / nansum(reference["areacello"])
        result["reference_height"] = reference_height
        result[variant] = reference_height * expansion_coeff
        result["masso"] = masso
        result["expansion_coeff"] = expansion_coeff
    else:  # steric.py:150-166
        delta_rho = np.where(
            ~np.isnan(reference["volcello"]), rho - reference["rho"], np.nan
        )
        result["delta_rho"] = delta_rho
        dz = calc_dz(z_l, z_i, deptho)
        sealevel = (-1.0 / rhozero) * nansum(dz * delta_rho, axis=1)
        wet = ~np.isnan(reference["volcello"][0])
        result[variant] = np.where(wet, sealevel, np.nan)
    return result, reference


# ----------------------------------------------------------------------------
# calendars / annual average (util.py:49-119) -- cftime is not installed, so the
# few calendar facts needed are restated here.
# ----------------------------------------------------------------------------
_DPM = (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31)


def is_leap(year, calendar):
    calendar = calendar.lower()
    if calendar in ("noleap", "365_day"):
        return False
    if calendar in ("all_leap", "366_day"):
        return True
    if calendar == "julian":
        return year % 4 == 0
    if calendar in ("standard", "gregorian", "proleptic_gregorian"):
        return (year % 4 == 0 and year % 100 != 0) or year % 400 == 0
    raise ValueError(f"unsupported calendar {calendar}")


def days_in_month(year, month, calendar):
    if calendar.lower() == "360_day":
        return 30
    d = _DPM[month - 1]
    if month == 2 and is_leap(year, calendar):
        d += 1
    return d


def monthly_time_axis(start_year, nyears, calendar):
    """Monthly mid-point axis of test_data/time.py:42-120 as (year, month, dim) rows."""
    rows = []
    for y in range(start_year, start_year + nyears):
        for m in range(1, 13):
            rows.append((y, m, days_in_month(y, m, calendar)))
    return rows


def annual_average(values, years, weights):
    """util.py:49-119 on a plain array: per-year days-in-month weighted mean.

    values: (nt, ...); years, weights: (nt,) ints.  Asserts 12 steps per year.
    xarray's weighted mean = sum(w*x, skipna) / sum(w where x notnull).
    """
    years = np.asarray(years)
    weights = np.asarray(weights, dtype=np.float64)
    out = []
    for yr in sorted(set(years.tolist())):
        sel = np.nonzero(years == yr)[0]
        assert len(sel) == 12
        x = values[sel]
        w = weights[sel].reshape((12,) + (1,) * (x.ndim - 1))
        num = np.sum(np.where(np.isnan(x), 0.0, x) * w, axis=0)
        den = np.sum(np.where(np.isnan(x), 0.0, 1.0) * w, axis=0)
        den = np.where(den != 0, den, np.nan)
        out.append(num / den)
    return np.stack(out, axis=0)


# ----------------------------------------------------------------------------
# test-data generator (config #1) -- src/momlevel/test_data/__init__.py:16-105,
# tripolar/horizontal.py:83-121, tripolar/vertical.py:37-84
# ----------------------------------------------------------------------------
def generate_test_data(start_year=1981, nyears=0, calendar="noleap", seed=123):
    """nt x 5 x 5 x 5 dataset as a dict of numpy arrays (nt=5, or 12*nyears)."""
    d = {}
    if nyears >= 1:
        axis = monthly_time_axis(start_year, nyears, calendar)
        d["time_year"] = np.array([r[0] for r in axis])
        d["time_month"] = np.array([r[1] for r in axis])
        d["time_days_in_month"] = np.array([r[2] for r in axis])
        ntimes = len(axis)
    else:
        d["time"] = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
        ntimes = 5
    d["calendar"] = calendar

    d["xh"] = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    d["yh"] = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    lon = np.arange(0.0, 361.0, 72.0)
    lat = np.arange(-90.0, 91.0, 36.0)
    lon = [(lon[x] + lon[x + 1]) / 2.0 for x in range(0, len(lon) - 1)]
    lat = [(lat[x] + lat[x + 1]) / 2.0 for x in range(0, len(lat) - 1)]
    d["geolon"], d["geolat"] = np.meshgrid(lon, lat)

    # xy_fields() and zlevel_fields() are called with their own default
    # seed=123 whatever the dataset seed is (test_data/__init__.py:63-64,
    # horizontal.py:11, vertical.py:13)
    areacello = np.random.default_rng(123).normal(100.0, 10.0, (5, 5))
    areacello = areacello / areacello.sum()
    d["areacello"] = areacello * 3.6111092e14

    d["z_i"] = np.array([0.0, 5.0, 15.0, 185.0, 1815.0, 6185.0])
    d["z_l"] = np.array([2.5, 10.0, 100.0, 1000.0, 4000.0])
    d["deptho"] = np.array(
        [
            np.random.default_rng(123).uniform(0.0, hi, 5)
            for hi in (5.0, 15.0, 185.0, 1815.0, 6185.0)
        ]
    )

    shape = (ntimes, 5, 5, 5)
    d["thetao"] = np.random.default_rng(seed).normal(15.0, 5.0, shape)
    d["so"] = np.random.default_rng(seed).normal(35.0, 1.5, shape)
    d["volcello"] = np.random.default_rng(seed).normal(1000.0, 100.0, shape)
    return d


def generate_test_data_dz(seed=123):
    """src/momlevel/test_data/__init__.py:108-140."""
    deptho = np.random.default_rng(seed).uniform(0.0, 100.0, (5, 5))
    deptho[2, 2] = np.nan
    deptho[2, 3] = np.nan
    z_i = np.array([0.0, 5.0, 10.0, 20.0, 50.0, 100.0])
    z_l = np.array((z_i[1::] + z_i[0:-1]) / 2.0)
    return {"deptho": deptho, "z_i": z_i, "z_l": z_l}


def generate_test_data_time(start_year=1981, nyears=5, calendar="noleap", seed=123):
    """src/momlevel/test_data/__init__.py:143-191, monthly ("MS") axis only."""
    axis = monthly_time_axis(start_year, nyears, calendar)
    nt = len(axis)
    return {
        "time_year": np.array([r[0] for r in axis]),
        "time_month": np.array([r[1] for r in axis]),
        "time_days_in_month": np.array([r[2] for r in axis]),
        "calendar": calendar,
        "var_a": np.random.default_rng(seed).normal(100, 20, (nt, 5, 5)),
        "var_b": np.random.default_rng(seed * 2).normal(100, 20, (nt, 5, 5)),
    }


# ----------------------------------------------------------------------------
# EXTENSION -- NOT a momlevel function.  PARITY UNPINNED: the reference has no ocean-heat-content
# code (grep heat|ohc over /root/reference/src is empty), so there is nothing to pin this against;
# it is the build's own definition (BASELINE.json configs[4] names the quantity), restated in numpy
# for the GPU tests of momlevel_amd.steric_variants(..., heat_content=True).
# ----------------------------------------------------------------------------
def ocean_heat_content(thetao, volcello_ref, rhozero=1035.0, cp=3992.0):
    """OHC(t) = rhozero * cp * sum_{z,y,x} thetao(t) * volcello_ref  [J, relative to 0 degC];
    skipna sum, float64 product (float32 theta is promoted by numpy)."""
    heat = nansum(thetao * volcello_ref, axis=(1, 2, 3))
    return (np.float64(rhozero) * np.float64(cp)) * heat
