// numpy's type promotion for the EOS functions: every dtype combination the reference accepts.
//
// The reference's EOS functions are bare numpy expressions (src/momlevel/eos/wright.py:44-48,
// 74-83, 108-117, 142, 165; eos/linear.py:44-46, 104, 124), so what they compute depends on the
// dtypes they are handed:
//   * float32 theta and salinity with a float64 pressure -- the steric path on MOM6 output -- keep
//     the polynomials in float32 and promote where the pressure enters (eos_device.hpp's
//     kF32Faithful kernels, the fast path);
//   * a PYTHON float pressure (calc_pdens: level*1e4 + patm, derived.py:477) or a float32 pressure
//     array keeps the WHOLE expression in float32;
//   * theta and salinity of different dtypes mix: each sub-expression has the dtype numpy's
//     promotion gives it;
//   * a python float for theta or salinity folds with the constants in float64 first.
// The rules (numpy >= 2, NEP 50): array (op) array -> the wider dtype; a python float is a WEAK
// scalar and takes the dtype of the array it meets; python float (op) python float is float64
// arithmetic and stays weak.  C++ arithmetic on float/double already IS the first rule (with
// -ffp-contract=off and no excess precision: float*float is rounded to float, float+double
// widens exactly), and the `Weak` type below adds the second and third.  So each function is
// written ONCE, operator for operator as the reference writes it, and instantiating it with
// TT, TS, TP in {float, double, Weak} gives all 27 dtype combinations bit for bit.
//
// Used by k_eos_promote (momlevel_promote.hip); the two combinations the steric path streams
// (all float64; float32 theta/S with float64 pressure) stay on the tuned kernels of
// momlevel_hip.hip, and tests/test_gpu_promote.py holds the two implementations against each
// other.  The header is plain C++14 and also compiles for the host (oracle/host_promote.cpp, the
// host build of the ABI).
#pragma once
#include <type_traits>

#if defined(__HIPCC__)
#define MLX_NP_FN __host__ __device__ __forceinline__
#else
#define MLX_NP_FN inline
#endif

namespace mlx {
namespace np {

struct Weak {  // a python float
  double v;
};
MLX_NP_FN constexpr Weak W(double v) { return Weak{v}; }

MLX_NP_FN Weak operator+(Weak a, Weak b) { return Weak{a.v + b.v}; }
MLX_NP_FN Weak operator-(Weak a, Weak b) { return Weak{a.v - b.v}; }
MLX_NP_FN Weak operator*(Weak a, Weak b) { return Weak{a.v * b.v}; }
MLX_NP_FN Weak operator/(Weak a, Weak b) { return Weak{a.v / b.v}; }

// python float (op) array element of dtype X: the scalar is converted to X first
template <typename X>
using IfElem = typename std::enable_if<std::is_floating_point<X>::value, X>::type;
template <typename X>
MLX_NP_FN IfElem<X> operator+(Weak a, X x) { return static_cast<X>(a.v) + x; }
template <typename X>
MLX_NP_FN IfElem<X> operator+(X x, Weak a) { return x + static_cast<X>(a.v); }
template <typename X>
MLX_NP_FN IfElem<X> operator-(Weak a, X x) { return static_cast<X>(a.v) - x; }
template <typename X>
MLX_NP_FN IfElem<X> operator-(X x, Weak a) { return x - static_cast<X>(a.v); }
template <typename X>
MLX_NP_FN IfElem<X> operator*(Weak a, X x) { return static_cast<X>(a.v) * x; }
template <typename X>
MLX_NP_FN IfElem<X> operator*(X x, Weak a) { return x * static_cast<X>(a.v); }
template <typename X>
MLX_NP_FN IfElem<X> operator/(Weak a, X x) { return static_cast<X>(a.v) / x; }
template <typename X>
MLX_NP_FN IfElem<X> operator/(X x, Weak a) { return x / static_cast<X>(a.v); }

MLX_NP_FN double widen(float x) { return (double)x; }  // exact
MLX_NP_FN double widen(double x) { return x; }
MLX_NP_FN double widen(Weak x) { return x.v; }

// is the numpy result of this type a float32 array?
template <typename R>
struct IsF32 : std::is_same<typename std::remove_cv<R>::type, float> {};

// np.full_like(T, c): T's dtype; of a python float, a float64 0-d array (a STRONG double)
template <typename TT>
struct FullLike {
  static MLX_NP_FN TT of(double c) { return static_cast<TT>(c); }
};
template <>
struct FullLike<Weak> {
  static MLX_NP_FN double of(double c) { return c; }
};

// ---- Wright (1997): src/momlevel/eos/wright.py:6-20 ---------------------------------------------
constexpr double A0 = 7.057924e-4, A1 = 3.480336e-7, A2 = -1.112733e-7;
constexpr double B0 = 5.790749e8, B1 = 3.516535e6, B2 = -4.002714e4, B3 = 2.084372e2,
                 B4 = 5.944068e5, B5 = -9.643486e3;
constexpr double C0 = 1.704853e5, C1 = 7.904722e2, C2 = -7.984422, C3 = 5.140652e-2,
                 C4 = -2.302158e2, C5 = -3.079464;

// eos/wright.py:44-46 (and :74-76, :108-110)
template <typename TT, typename TS>
MLX_NP_FN auto wright_al0(TT T, TS S) {
  return W(A0) + W(A1) * T + W(A2) * S;
}
template <typename TT, typename TS>
MLX_NP_FN auto wright_p0(TT T, TS S) {
  return W(B0) + W(B4) * S + T * (W(B1) + T * (W(B2) + W(B3) * T) + W(B5) * S);
}
template <typename TT, typename TS>
MLX_NP_FN auto wright_lam(TT T, TS S) {
  return W(C0) + W(C4) * S + T * (W(C1) + T * (W(C2) + W(C3) * T) + W(C5) * S);
}

// eos/wright.py:47-48
template <typename TT, typename TS, typename TP>
MLX_NP_FN auto wright_density(TT T, TS S, TP p) {
  const auto al0 = wright_al0(T, S);
  const auto p0 = wright_p0(T, S);
  const auto lam = wright_lam(T, S);
  const auto I_denom = W(1.0) / (lam + al0 * (p + p0));
  return (p + p0) * I_denom;
}

// eos/wright.py:78-83
template <typename TT, typename TS, typename TP>
MLX_NP_FN auto wright_drho_dtemp(TT T, TS S, TP p) {
  const auto al0 = wright_al0(T, S);
  const auto p0 = wright_p0(T, S);
  const auto lam = wright_lam(T, S);
  auto I_denom2 = W(1.0) / (lam + al0 * (p + p0));
  I_denom2 = I_denom2 * I_denom2;
  return I_denom2 *
         (lam * (W(B1) + T * (W(2.0) * W(B2) + W(3.0) * W(B3) * T) + W(B5) * S) -
          (p + p0) * ((p + p0) * W(A1) +
                      (W(C1) + T * (W(C2) * W(2.0) + W(C3) * W(3.0) * T) + W(C5) * S)));
}

// eos/wright.py:112-117
template <typename TT, typename TS, typename TP>
MLX_NP_FN auto wright_drho_dsal(TT T, TS S, TP p) {
  const auto al0 = wright_al0(T, S);
  const auto p0 = wright_p0(T, S);
  const auto lam = wright_lam(T, S);
  auto I_denom2 = W(1.0) / (lam + al0 * (p + p0));
  I_denom2 = I_denom2 * I_denom2;
  return I_denom2 *
         (lam * (W(B4) + W(B5) * T) - (p + p0) * ((p + p0) * W(A2) + (W(C4) + W(C5) * T)));
}

// eos/wright.py:142
template <typename TT, typename TS, typename TP>
MLX_NP_FN auto wright_alpha(TT T, TS S, TP p) {
  return W(-1.0) * (wright_drho_dtemp(T, S, p) / wright_density(T, S, p));
}

// eos/wright.py:165
template <typename TT, typename TS, typename TP>
MLX_NP_FN auto wright_beta(TT T, TS S, TP p) {
  return wright_drho_dsal(T, S, p) / wright_density(T, S, p);
}

// ---- linear EOS: src/momlevel/eos/linear.py:13-19, 44-46 (rho_ref=None), 104, 124 -----------------
constexpr double RHO_T0_S0 = 1000.0, DRHO_DT = -0.2, DRHO_DS = 0.8;

template <typename TT, typename TS>
MLX_NP_FN auto linear_density(TT T, TS S) {
  return W(RHO_T0_S0) + ((W(DRHO_DT) * T) + (W(DRHO_DS) * S));
}
// eos/linear.py:55-56 with rho_ref given: c = RHO_T0_S0 - rho_ref (formed by the caller, in
// python: a float, or a numpy scalar when rho_ref is one), then c + ((DRHO_DT * T) + (DRHO_DS * S))
template <typename TT, typename TS, typename TC>
MLX_NP_FN auto linear_density_ref(TT T, TS S, TC c) {
  return c + ((W(DRHO_DT) * T) + (W(DRHO_DS) * S));
}
template <typename TT, typename TS>
MLX_NP_FN auto linear_alpha(TT T, TS S) {
  return W(-1.0) * (FullLike<TT>::of(DRHO_DT) / linear_density(T, S));
}
template <typename TT, typename TS>
MLX_NP_FN auto linear_beta(TT T, TS S) {
  return FullLike<TT>::of(DRHO_DS) / linear_density(T, S);
}

// ---- dispatch by (eos, func): the values of MLX_EOS_* / MLX_FUNC_* -------------------------------
constexpr int kEosWright = 0, kEosLinear = 1;
constexpr int kFnDensity = 0, kFnDrhoDtemp = 1, kFnDrhoDsal = 2, kFnAlpha = 3, kFnBeta = 4,
              kFnIbh = 5, kFnDensityRef = 6;

// One cell.  The result is returned widened to double (exact) and *is_f32 says whether numpy's
// result dtype is float32.  dynamic.py:34-36: ibh = pso * (-1.0 / (rho_conv * gravity)), gravity a
// python float.  The linear EOS's drho_dtemp / drho_dsal return the python constants themselves.
template <typename TT, typename TS, typename TP>
MLX_NP_FN double eval(int eos, int func, TT T, TS S, TP p, double gravity, bool* is_f32) {
#define MLX_NP_RETURN(expr)                    \
  do {                                         \
    const auto r_ = (expr);                    \
    *is_f32 = IsF32<decltype(r_)>::value;      \
    return widen(r_);                          \
  } while (0)
  if (eos == kEosLinear) {
    switch (func) {
      case kFnDensity: MLX_NP_RETURN(linear_density(T, S));
      case kFnDrhoDtemp: MLX_NP_RETURN(W(DRHO_DT));
      case kFnDrhoDsal: MLX_NP_RETURN(W(DRHO_DS));
      case kFnAlpha: MLX_NP_RETURN(linear_alpha(T, S));
      case kFnBeta: MLX_NP_RETURN(linear_beta(T, S));
      case kFnDensityRef: MLX_NP_RETURN(linear_density_ref(T, S, p));  // (p carries the constant)
      default: MLX_NP_RETURN(p * (W(-1.0) / (linear_density(T, S) * W(gravity))));
    }
  }
  switch (func) {
    case kFnDensity: MLX_NP_RETURN(wright_density(T, S, p));
    case kFnDrhoDtemp: MLX_NP_RETURN(wright_drho_dtemp(T, S, p));
    case kFnDrhoDsal: MLX_NP_RETURN(wright_drho_dsal(T, S, p));
    case kFnAlpha: MLX_NP_RETURN(wright_alpha(T, S, p));
    case kFnBeta: MLX_NP_RETURN(wright_beta(T, S, p));
    default: MLX_NP_RETURN(p * (W(-1.0) / (wright_density(T, S, p) * W(gravity))));
  }
#undef MLX_NP_RETURN
}

}  // namespace np
}  // namespace mlx
