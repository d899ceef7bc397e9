""" linear.py -- linear equation of state on the MI355X

Mirror of src/momlevel/eos/linear.py:17-162: ``density`` (rho = 1000 - 0.2 T + 0.8 S), the
constant derivatives ``drho_dtemp`` / ``drho_dsal`` and ``alpha`` / ``beta`` (the constants
divided by the density).  ``equation_of_state="linear"`` reaches the steric entry points and
``derived.calc_rho / calc_alpha / calc_beta`` through these; the arrays are evaluated by the
same HIP kernels as the Wright EOS (eos id MLX_EOS_LINEAR), bit-identical to numpy.
"""

from ._dispatch import evaluate

__all__ = ["alpha", "beta", "density", "drho_dsal", "drho_dtemp"]

RHO_REF = 1035.0
RHO_T0_S0 = 1000.0
DRHO_DT = -0.2
DRHO_DS = 0.8


def density(T, S, p=None, rho_ref=None):
    """In-situ density of the linear EOS; pressure is ignored (eos/linear.py:26-58)."""
    if rho_ref is None:
        return evaluate("linear", "density", T, S, None)
    # eos/linear.py:55-56: the constant term is formed first, in python -- a float, or a numpy scalar
    # when rho_ref is one -- and then meets the arrays: c + ((DRHO_DT * T) + (DRHO_DS * S)); the
    # kernel takes c where the other functions take the pressure (MLX_FUNC_DENSITY_REF)
    return evaluate("linear", "density_ref", T, S, RHO_T0_S0 - rho_ref)


def drho_dtemp(T=None, S=None, p=None):
    """d(rho)/d(theta) of the linear EOS: a constant, whatever T, S, p (eos/linear.py:61-84)."""
    return DRHO_DT


def drho_dsal(T=None, S=None, p=None):
    """d(rho)/d(S) of the linear EOS: a constant (eos/linear.py:87-110)."""
    return DRHO_DS


def alpha(T, S, p=None):
    """Thermal expansion coefficient -1.0 * (DRHO_DT / density) (eos/linear.py:113-136)."""
    return evaluate("linear", "alpha", T, S, None)


def beta(T, S, p=None):
    """Haline contraction coefficient DRHO_DS / density (eos/linear.py:139-162)."""
    return evaluate("linear", "beta", T, S, None)
