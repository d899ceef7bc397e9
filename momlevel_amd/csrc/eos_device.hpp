// eos_device.hpp -- pointwise equations of state, device side (gfx950).
//
// The arithmetic follows the reference's numpy expressions OPERATOR FOR OPERATOR
// (src/momlevel/eos/wright.py:44-48, 74-83, 108-117, 142, 165 and
// src/momlevel/eos/linear.py:55-56), with floating-point contraction switched
// off, so that every finite result is bit-identical to what numpy computes on
// the host: each + - * / below is one correctly rounded IEEE-754 operation,
// evaluated in the order Python's precedence rules give the reference's source.
// The f64 division is hipcc's IEEE-compliant v_div_scale/v_rcp/v_fma/v_div_fmas/
// v_div_fixup sequence (correctly rounded).
//
// This file must be compiled with -ffp-contract=off (csrc/build.py does); the
// pragma below is a second line of defence.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace mlx {

// dtype modes of the streamed theta/S fields (mirror MLX_DTYPE_* in momlevel_hip.h)
constexpr int kF64 = 0;
constexpr int kF32Faithful = 1;  // numpy's mixed precision for float32 theta/S
constexpr int kF32Upcast = 2;    // upcast to float64 first

constexpr int kWright = 0;
constexpr int kLinear = 1;

constexpr int kDensity = 0;
constexpr int kDrhoDtemp = 1;
constexpr int kDrhoDsal = 2;
constexpr int kAlpha = 3;
constexpr int kBeta = 4;
constexpr int kIbh = 5;  // inverse barometer height (dynamic.py:34-36); aux = gravity

// src/momlevel/eos/wright.py:6-20
template <typename R>
struct WrightC {
  static constexpr R A0 = R(7.057924e-4);
  static constexpr R A1 = R(3.480336e-7);
  static constexpr R A2 = R(-1.112733e-7);
  static constexpr R B0 = R(5.790749e8);
  static constexpr R B1 = R(3.516535e6);
  static constexpr R B2 = R(-4.002714e4);
  static constexpr R B3 = R(2.084372e2);
  static constexpr R B4 = R(5.944068e5);
  static constexpr R B5 = R(-9.643486e3);
  static constexpr R C0 = R(1.704853e5);
  static constexpr R C1 = R(7.904722e2);
  static constexpr R C2 = R(-7.984422);
  static constexpr R C3 = R(5.140652e-2);
  static constexpr R C4 = R(-2.302158e2);
  static constexpr R C5 = R(-3.079464);
};

// al0, p0, lam in precision R (double, or float for numpy's float32 inputs: the
// python-float constants are weak scalars and are rounded to float32 first)
template <typename R>
__device__ __forceinline__ void wright_terms(R T, R S, R& al0, R& p0, R& lam) {
  using K = WrightC<R>;
  al0 = (K::A0 + K::A1 * T) + K::A2 * S;
  p0 = (K::B0 + K::B4 * S) + T * ((K::B1 + T * (K::B2 + K::B3 * T)) + K::B5 * S);
  lam = (K::C0 + K::C4 * S) + T * ((K::C1 + T * (K::C2 + K::C3 * T)) + K::C5 * S);
}

// ---- held-field hoisting (thermosteric: S fixed in time; halosteric: T fixed) --------------
// The sub-expressions of al0, p0, lam that depend only on the held field are evaluated once
// per cell, outside the time loop.  They are the SAME sub-expressions, rounded the same way,
// so the result stays bit-identical to wright_terms(); only 7 (S held) or 10 (T held) of the
// ~26 polynomial operations per cell and time step disappear.
template <typename R>
struct HeldS {  // terms of S only
  R a2s, b04s, b5s, c04s, c5s;
};
template <typename R>
struct HeldT {  // terms of T only
  R t, a01t, bpoly, cpoly;
};

template <typename R>
__device__ __forceinline__ HeldS<R> hold_S(R S) {
  using K = WrightC<R>;
  HeldS<R> h;
  h.a2s = K::A2 * S;
  h.b04s = K::B0 + K::B4 * S;
  h.b5s = K::B5 * S;
  h.c04s = K::C0 + K::C4 * S;
  h.c5s = K::C5 * S;
  return h;
}

template <typename R>
__device__ __forceinline__ HeldT<R> hold_T(R T) {
  using K = WrightC<R>;
  HeldT<R> h;
  h.t = T;
  h.a01t = K::A0 + K::A1 * T;
  h.bpoly = K::B1 + T * (K::B2 + K::B3 * T);
  h.cpoly = K::C1 + T * (K::C2 + K::C3 * T);
  return h;
}

template <typename R>
__device__ __forceinline__ void wright_terms_heldS(R T, const HeldS<R>& h, R& al0, R& p0, R& lam) {
  using K = WrightC<R>;
  al0 = (K::A0 + K::A1 * T) + h.a2s;
  p0 = h.b04s + T * ((K::B1 + T * (K::B2 + K::B3 * T)) + h.b5s);
  lam = h.c04s + T * ((K::C1 + T * (K::C2 + K::C3 * T)) + h.c5s);
}

template <typename R>
__device__ __forceinline__ void wright_terms_heldT(const HeldT<R>& h, R S, R& al0, R& p0, R& lam) {
  using K = WrightC<R>;
  al0 = h.a01t + K::A2 * S;
  p0 = (K::B0 + K::B4 * S) + h.t * (h.bpoly + K::B5 * S);
  lam = (K::C0 + K::C4 * S) + h.t * (h.cpoly + K::C5 * S);
}

// rho from (al0, p0, lam) already widened to float64: eos/wright.py:47-48
__device__ __forceinline__ double wright_density_from_terms(double al0, double p0, double lam,
                                                            double p) {
  const double pp0 = p + p0;
  const double I_denom = 1.0 / (lam + al0 * pp0);
  return pp0 * I_denom;
}

// arithmetic type of the polynomial part for a dtype mode
template <int MODE>
struct PolyType {
  typedef double type;
};
template <>
struct PolyType<kF32Faithful> {
  typedef float type;
};

// in-situ density, eos/wright.py:44-48.  MODE selects how float32 inputs are treated.
template <int MODE, typename TIn>
__device__ __forceinline__ double wright_density(TIn Tin, TIn Sin, double p) {
  double al0, p0, lam;
  if constexpr (MODE == kF32Faithful) {
    float a, b, c;
    wright_terms<float>(Tin, Sin, a, b, c);
    al0 = (double)a;
    p0 = (double)b;
    lam = (double)c;
  } else {
    wright_terms<double>((double)Tin, (double)Sin, al0, p0, lam);
  }
  const double pp0 = p + p0;
  const double I_denom = 1.0 / (lam + al0 * pp0);
  return pp0 * I_denom;
}

// eos/wright.py:74-83 (float64 only)
__device__ __forceinline__ double wright_drho_dtemp(double T, double S, double p) {
  using K = WrightC<double>;
  double al0, p0, lam;
  wright_terms<double>(T, S, al0, p0, lam);
  const double pp0 = p + p0;
  double I2 = 1.0 / (lam + al0 * pp0);
  I2 = I2 * I2;
  const double a = lam * ((K::B1 + T * (2.0 * K::B2 + (3.0 * K::B3) * T)) + K::B5 * S);
  const double b =
      pp0 * (pp0 * K::A1 + ((K::C1 + T * (K::C2 * 2.0 + (K::C3 * 3.0) * T)) + K::C5 * S));
  return I2 * (a - b);
}

// eos/wright.py:108-117
__device__ __forceinline__ double wright_drho_dsal(double T, double S, double p) {
  using K = WrightC<double>;
  double al0, p0, lam;
  wright_terms<double>(T, S, al0, p0, lam);
  const double pp0 = p + p0;
  double I2 = 1.0 / (lam + al0 * pp0);
  I2 = I2 * I2;
  return I2 * (lam * (K::B4 + K::B5 * T) - pp0 * (pp0 * K::A2 + (K::C4 + K::C5 * T)));
}

// ---- float32 theta/S, numpy's mixed precision (MODE kF32Faithful) for the derivatives ---------
// With float32 arrays and python-float constants numpy keeps every sub-expression that involves
// only T, S and constants in float32, and promotes to float64 wherever the float64 pressure
// enters.  Written out operator by operator for eos/wright.py:74-83 and :108-117:
__device__ __forceinline__ double wright_drho_dtemp_f32(float T, float S, double p) {
  using K = WrightC<float>;
  float al0, p0, lam;
  wright_terms<float>(T, S, al0, p0, lam);
  const double pp0 = p + (double)p0;
  double I2 = 1.0 / ((double)lam + (double)al0 * pp0);
  I2 = I2 * I2;
  // lam * (B1 + T*(2.0*B2 + 3.0*B3*T) + B5*S): float32 throughout (2.0*B2 and 3.0*B3 are python
  // floats folded in float64 first, then rounded to float32 when they meet the array)
  const float two_b2 = (float)(2.0 * WrightC<double>::B2), three_b3 = (float)(3.0 * WrightC<double>::B3);
  const float two_c2 = (float)(WrightC<double>::C2 * 2.0), three_c3 = (float)(WrightC<double>::C3 * 3.0);
  const float a = lam * ((K::B1 + T * (two_b2 + three_b3 * T)) + K::B5 * S);
  const float cpoly = (K::C1 + T * (two_c2 + three_c3 * T)) + K::C5 * S;
  const double b = pp0 * (pp0 * WrightC<double>::A1 + (double)cpoly);
  return I2 * ((double)a - b);
}

__device__ __forceinline__ double wright_drho_dsal_f32(float T, float S, double p) {
  using K = WrightC<float>;
  float al0, p0, lam;
  wright_terms<float>(T, S, al0, p0, lam);
  const double pp0 = p + (double)p0;
  double I2 = 1.0 / ((double)lam + (double)al0 * pp0);
  I2 = I2 * I2;
  const float a = lam * (K::B4 + K::B5 * T);
  const float c = K::C4 + K::C5 * T;
  const double b = pp0 * (pp0 * WrightC<double>::A2 + (double)c);
  return I2 * ((double)a - b);
}

// eos/linear.py:55-56 with rho_ref=None: 1000 + ((-0.2*T) + (0.8*S))
template <int MODE, typename TIn>
__device__ __forceinline__ double linear_density(TIn Tin, TIn Sin) {
  if constexpr (MODE == kF32Faithful) {
    // numpy: float32 array * weak python float stays float32; 1000.0 + f32 -> f32
    const float r = 1000.0f + ((-0.2f * Tin) + (0.8f * Sin));
    return (double)r;
  } else {
    const double T = (double)Tin, S = (double)Sin;
    return 1000.0 + ((-0.2 * T) + (0.8 * S));
  }
}

// runtime-dispatched EOS function (generic kernels; eos/func are wave-uniform)
template <int MODE, typename TIn>
__device__ __forceinline__ double eos_eval(int eos, int func, TIn T, TIn S, double p,
                                           double aux = 0.0) {
  if (func == kIbh) {  // pso * (-1.0 / (rho_conv * gravity))
    const double rho = (eos == kLinear) ? linear_density<MODE, TIn>(T, S)
                                        : wright_density<MODE, TIn>(T, S, p);
    return p * (-1.0 / (rho * aux));
  }
  if (eos == kLinear) {
    return linear_density<MODE, TIn>(T, S);
  }
  switch (func) {
    case kDensity:
      return wright_density<MODE, TIn>(T, S, p);
    case kDrhoDtemp:
      if constexpr (MODE == kF32Faithful) return wright_drho_dtemp_f32(T, S, p);
      else return wright_drho_dtemp((double)T, (double)S, p);
    case kDrhoDsal:
      if constexpr (MODE == kF32Faithful) return wright_drho_dsal_f32(T, S, p);
      else return wright_drho_dsal((double)T, (double)S, p);
    case kAlpha:  // eos/wright.py:142
      if constexpr (MODE == kF32Faithful)
        return -1.0 * (wright_drho_dtemp_f32(T, S, p) / wright_density<MODE, TIn>(T, S, p));
      else
        return -1.0 * (wright_drho_dtemp((double)T, (double)S, p) /
                       wright_density<kF64, double>((double)T, (double)S, p));
    default:  // kBeta, eos/wright.py:165
      if constexpr (MODE == kF32Faithful)
        return wright_drho_dsal_f32(T, S, p) / wright_density<MODE, TIn>(T, S, p);
      else
        return wright_drho_dsal((double)T, (double)S, p) /
               wright_density<kF64, double>((double)T, (double)S, p);
  }
}

__device__ __forceinline__ bool is_nan(double x) { return x != x; }

__device__ __forceinline__ double canonical_nan() {
  return __longlong_as_double(0x7FF8000000000000LL);
}

// splitmix64 -- synthetic-field generator; replayed in numpy by
// momlevel_amd/synthetic.py
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  unsigned long long z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

}  // namespace mlx
