""" dynamic.py - inverse barometer height on the MI355X (src/momlevel/dynamic.py:8-41)

One fused EOS call: ``ibh = pso * (-1 / (rho(tos, sos, pso) * gravity))`` is evaluated inside the
EOS kernel (mlx_inverse_barometer), operator for operator as the reference writes it.
"""

from . import util
from .adapters import accepts_xarray
from .derived import _apply_eos
from .eos._dispatch import evaluate

__all__ = ["inverse_barometer"]


@accepts_xarray
def inverse_barometer(tos, sos, pso, gravity=9.8, equation_of_state="Wright"):
    """Inverse barometer height in m from sea surface temperature, salinity and pressure."""
    util.eos_func_from_str(equation_of_state)  # unknown EOS -> ValueError

    def fused(T, S, p):
        return evaluate(equation_of_state.lower(), "inverse_barometer", T, S, p, gravity=gravity)

    ibh = _apply_eos("density", tos, sos, pso, equation_of_state, eos_func=fused)
    ibh = ibh.rename("ibh")
    ibh.attrs = {"long_name": "Inverse Barometer Height", "units": "m"}
    return ibh
