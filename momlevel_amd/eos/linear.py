""" linear.py -- linear equation of state on the MI355X

``density`` of src/momlevel/eos/linear.py:26-58 (rho = 1000 - 0.2 T + 0.8 S), kept so
that ``equation_of_state="linear"`` stays reachable through the steric entry points.
It runs in the same HIP kernels as the Wright EOS (eos id MLX_EOS_LINEAR); the
constant-derivative helpers of the reference's linear module are not part of the
steric path and are not provided.
"""

from ._dispatch import evaluate

__all__ = ["density"]

RHO_REF = 1035.0
RHO_T0_S0 = 1000.0
DRHO_DT = -0.2
DRHO_DS = 0.8


def density(T, S, p=None, rho_ref=None):
    """In-situ density of the linear EOS; pressure is ignored (eos/linear.py:26-58)."""
    rho = evaluate("linear", "density", T, S, None)
    if rho_ref is not None:
        # reference: rho = (1000 - rho_ref) + (...); only the rho_ref=None form is bit-exact here
        rho = rho - rho_ref
    return rho
