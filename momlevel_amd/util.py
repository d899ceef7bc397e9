"""Host-side helpers of the steric path: the reference's ``momlevel.util`` hot subset.

Same names, argument meaning and error behaviour as src/momlevel/util.py
(default_coords :199-224, eos_func_from_str :227-249, validate_areacello
:669-694, validate_dataset :697-814, annual_average :49-119).  No device work
except the area sum of device-resident ``areacello`` (mlx_nansum).
"""

import inspect
import warnings

import numpy as np

from . import eos
from .labeled import DataArray, Dataset

__all__ = [
    "annual_average",
    "default_coords",
    "eos_func_from_str",
    "validate_areacello",
    "validate_dataset",
]


_COORD_DEFAULTS = (("t", "time"), ("z", "z_l"), ("zbounds", "z_i"))


def default_coords(coord_names=None):
    """``(tcoord, zcoord, zbounds)``: MOM6's ``("time", "z_l", "z_i")`` unless the mapping
    overrides them under the keys ``"t"``, ``"z"``, ``"zbounds"`` (behaviour of util.py:199-224;
    a non-dict mapping is an ``AssertionError`` there, and here)."""
    mapping = coord_names if coord_names is not None else {}
    if not isinstance(mapping, dict):
        raise AssertionError("Coordinate mapping must be a dictionary.")
    return tuple(mapping.get(key, fallback) for key, fallback in _COORD_DEFAULTS)


def eos_func_from_str(eos_str, func_name="density"):
    """The function ``func_name`` of the equation-of-state module named ``eos_str`` (any case):
    ``"Wright"`` -> ``momlevel_amd.eos.wright.density``.  Unknown equation of state ->
    ``ValueError("Unknown equation of state: ...")`` as util.py:227-249; a function the module does
    not provide (the linear EOS has no alpha / beta here) is a ``ValueError`` too, not the bare
    ``KeyError`` of the reference's ``__dict__`` lookup."""
    if not isinstance(eos_str, str):
        raise AssertionError("Expecting string for equation of state")
    name = eos_str.lower()
    module = None if name.startswith("_") else getattr(eos, name, None)
    if not inspect.ismodule(module):
        raise ValueError(f"Unknown equation of state: {name}")
    func = getattr(module, func_name, None)
    if not callable(func):
        raise ValueError(
            f"Equation of state '{name}' does not provide `{func_name}` in momlevel_amd"
        )
    return func


def _area_sum(areacello):
    """skipna sum of areacello; on the device (mlx_nansum) when the data lives there."""
    if isinstance(areacello, DataArray) and areacello.is_device:
        from . import core

        return float(core.nansum(areacello.data).item())
    return float(np.nansum(np.asarray(areacello)))


def validate_areacello(areacello, reference=3.6111092e14, tolerance=0.02, total=None):
    """Does ``areacello`` sum to the real ocean's surface (3.6111092e14 m2) within a relative
    ``tolerance``?  Catches an unmasked field, e.g. the whole globe's area (util.py:669-694).
    ``total`` (not in momlevel): the sum over ALL horizontal tiles when ``areacello`` is one rank's
    tile of a multi-GPU run (momlevel_amd.parallel) -- the check is about the whole ocean."""
    area = _area_sum(areacello) if total is None else float(total)
    relative_error = (area - reference) / reference
    return bool(abs(relative_error) < tolerance)


_FIELDS_4D = ("thetao", "so", "volcello")
_FIELDS_2D = ("areacello", "deptho")
_REFERENCE_SCALARS = ("masso", "volo", "rhoga")
_REQUIRED = ("thetao", "so", "volcello", "areacello")
_REQUIRED_IN_REFERENCE = ("rho", "volo", "masso", "rhoga")


def _dataset_findings(dset, reference, additional_vars, area_total=None):
    """Yield ``(severity, message)`` for every requirement ``dset`` does not meet, in the order
    util.py:697-814 reports them.  severity "area" marks the one finding ``strict=False`` turns
    into a warning."""
    present = set(dset.variables)
    extra = [] if additional_vars is None else (
        list(additional_vars) if isinstance(additional_vars, list) else [additional_vars])
    wanted = list(_REQUIRED) + extra + (list(_REQUIRED_IN_REFERENCE) if reference else [])
    missing = [name for name in dict.fromkeys(wanted) if name not in present]
    if missing:
        yield "error", f"Reference dataset is missing variables: {missing}"

    # (a reference dataset "cannot contain a time coordinate": the reference's test compares names
    #  with unbound ``str.lower`` methods and can never fire, util.py:728-733 -- nothing to report)

    nd, layout = (3, "(z,y,x)") if reference else (4, "t,z,y,x")
    rank_rules = [(name, nd, f"Variable {name} must have exactly {nd} dimensions {layout}")
                  for name in _FIELDS_4D]
    rank_rules += [(name, 2, f"Variable {name} must have exactly 2 dimensions (y,x)")
                   for name in _FIELDS_2D]
    for name, rank, message in rank_rules:
        if name in present and len(dset[name].dims) != rank:
            yield "error", message

    if "areacello" in present and not validate_areacello(dset["areacello"], total=area_total):
        yield "area", "Variable `areacello` field is out of range. It may not be masked."

    if reference:
        # the reference words the `rho` finding with the name `areacello` (util.py:800);
        # the text is what a user of either package sees, so it is kept
        if "rho" in present and len(dset["rho"].dims) != 3:
            yield "error", "Variable areacello must have exactly 3 dimensions (z,y,x)"
        for name in _REFERENCE_SCALARS:
            if name in present and len(dset[name].dims) != 0:
                yield "error", f"Variable {name} must be a scalar"


def local_findings(dset, reference=False, additional_vars=None):
    """The fatal findings that do not depend on the areacello range check -- what ONE RANK of a
    tiled run can establish about its own tile before the first collective (steric.all_ranks_ok)."""
    # area_total=reference value: the range finding cannot fire here; it is judged on the global sum
    return [msg for severity, msg in _dataset_findings(dset, reference, additional_vars,
                                                       area_total=3.6111092e14)
            if severity != "area"]


def validate_dataset(dset, reference=False, strict=True, additional_vars=None, area_total=None):
    """Is ``dset`` a usable input (or, ``reference=True``, reference-state) dataset?

    Same observable behaviour as util.py:697-814: required variables present, fields of the right
    rank, ``areacello`` summing to the real ocean's area.  All findings are printed, then ONE
    ``ValueError("Errors found in dataset.")`` is raised; ``strict=False`` downgrades only the
    areacello range finding to a ``UserWarning``.  Returns None.  ``area_total``: see
    validate_areacello (tiled multi-GPU runs only).
    """
    fatal = []
    for severity, message in _dataset_findings(dset, reference, additional_vars, area_total):
        if severity == "area" and not strict:
            warnings.warn(message)
        else:
            fatal.append(message)
    if fatal:
        print("\n".join(fatal))
        raise ValueError("Errors found in dataset.")


# ---------------------------------------------------------------------------------------
# annual_average (util.py:49-119).  The time coordinate must hold calendar-aware
# objects exposing .year, .month, .calendar and .daysinmonth -- cftime.datetime does;
# momlevel_amd.cftime_lite.DatetimeLite is the stand-in when cftime is not installed.
# ---------------------------------------------------------------------------------------
def _annual_weights(time_values):
    years = np.array([t.year for t in time_values])
    weights = np.array([float(t.daysinmonth) for t in time_values])
    return years, weights


def _annual_mean_array(values, years, weights):
    """Per-year weighted mean over axis 0 (xarray ``weighted(w).mean``: NaNs carry no weight)."""
    out = []
    for yr in sorted(set(years.tolist())):
        sel = np.nonzero(years == yr)[0]
        assert len(sel) == 12
        x = values[sel]
        w = weights[sel].reshape((12,) + (1,) * (x.ndim - 1))
        num = np.sum(np.where(np.isnan(x), 0.0, x) * w, axis=0)
        den = np.sum(np.where(np.isnan(x), 0.0, 1.0) * w, axis=0)
        out.append(num / np.where(den != 0, den, np.nan))
    return np.stack(out, axis=0)


def _np_dtype(var):
    """numpy dtype of a labelled variable WITHOUT copying device data to the host."""
    from .labeled import np_dtype

    return np_dtype(var.data.dtype)


class AnnualPlan:
    """Calendar bookkeeping of annual_average for one time coordinate."""

    def __init__(self, time_da, tcoord):
        from .cftime_lite import year_midpoint

        time_values = list(time_da.values)
        calendar = time_values[0].calendar
        self.years, self.weights = _annual_weights(time_values)
        self.year_list = sorted(set(self.years.tolist()))
        for yr in self.year_list:
            assert int(np.sum(self.years == yr)) == 12  # util.py:85
        new_time = [year_midpoint(int(y), calendar) for y in self.year_list]
        self.time = DataArray(np.array(new_time, dtype=object), (tcoord,), None, time_da.attrs,
                              tcoord)
        # whole years stored back to back in ascending order: the device kernel's layout
        self.contiguous = bool(np.array_equal(self.years, np.repeat(self.year_list, 12)))


def annual_average(xobj, tcoord="time"):
    """Days-in-month weighted annual means (util.py:49-119).

    Accepts a labelled Dataset or DataArray; asserts 12 steps per year; the new
    time axis holds the mid-points of the years.
    """
    plan = AnnualPlan(xobj[tcoord], tcoord)
    years, weights, time_da = plan.years, plan.weights, plan.time

    def avg(da):
        if tcoord not in da.dims:
            return da
        moved = da.transpose(tcoord, ...)
        if moved.is_device and plan.contiguous:  # device-resident data: mlx_group_weighted_mean
            from . import core

            res = core.group_weighted_mean(moved.data.contiguous(), weights, 12)
        else:
            res = _annual_mean_array(moved.values.astype(np.float64), years, weights)
        coords = {k: v for k, v in moved.coords.items() if tcoord not in v.dims}
        coords[tcoord] = time_da
        out = DataArray(res, moved.dims, coords, da.attrs, da.name)
        out.encoding = dict(da.encoding)
        return out

    if isinstance(xobj, Dataset):
        result = Dataset(attrs=xobj.attrs)
        for name, c in xobj.coords.items():
            if tcoord not in c.dims:
                result._set(name, c, is_coord=True)
        result._set(tcoord, time_da, is_coord=True)
        for name, var in xobj.data_vars.items():
            if np.dtype(_np_dtype(var)).kind not in "fiu":
                continue  # non-numeric variables are skipped (util.py:79-84)
            result[name] = avg(var)
        return result
    return avg(xobj)
