""" wright.py -- Wright (1997) equation of state on the MI355X

Drop-in for src/momlevel/eos/wright.py: the same five functions with the same
``(T, S, p)`` signature and numpy broadcasting.  The arithmetic runs in the HIP
kernel ``k_eos_map`` (csrc/momlevel_hip.hip), which keeps the reference's operator
order with FMA contraction off: finite float64 results are bit-identical to the
reference's numpy evaluation.
"""

from ._dispatch import evaluate

__all__ = ["density", "drho_dtemp", "drho_dsal", "alpha", "beta"]

# constants (src/momlevel/eos/wright.py:6-20); the device copy lives in csrc/eos_device.hpp
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


def density(T, S, p):
    """In-situ density (kg m-3) from potential temperature (degC), salinity (PSU)
    and absolute pressure (Pa).  Wright, 1997, J. Atmos. Ocean. Tech., 14, 735-740.
    (src/momlevel/eos/wright.py:23-50)"""
    return evaluate("wright", "density", T, S, p)


def drho_dtemp(T, S, p):
    """d(rho)/d(theta) in kg m-3 degC-1 (src/momlevel/eos/wright.py:53-85)"""
    return evaluate("wright", "drho_dtemp", T, S, p)


def drho_dsal(T, S, p):
    """d(rho)/d(S) in kg m-3 PSU-1 (src/momlevel/eos/wright.py:88-119)"""
    return evaluate("wright", "drho_dsal", T, S, p)


def alpha(T, S, p):
    """Thermal expansion coefficient in degC-1 (src/momlevel/eos/wright.py:122-142)"""
    return evaluate("wright", "alpha", T, S, p)


def beta(T, S, p):
    """Haline contraction coefficient in PSU-1 (src/momlevel/eos/wright.py:145-165)"""
    return evaluate("wright", "beta", T, S, p)
