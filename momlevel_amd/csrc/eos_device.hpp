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

#include <type_traits>

#pragma clang fp contract(off)

namespace mlx {

// dtype modes of the streamed theta/S fields (mirror MLX_DTYPE_* in momlevel_hip.h)
constexpr int kF64 = 0;
constexpr int kF32Faithful = 1;  // numpy's mixed precision for float32 theta/S
constexpr int kF32Upcast = 2;    // upcast to float64 first
// theta and salinity of DIFFERENT dtypes (generic kernels only).  numpy evaluates each part of the
// polynomial in its own field's precision and promotes to float64 where the two meet: see
// wright_density_mixed below
constexpr int kMixT32 = 3;       // theta float32, salinity float64
constexpr int kMixS32 = 4;       // theta float64, salinity float32
template <int MODE>
struct IsMixed {
  static constexpr bool value = (MODE == kMixT32 || MODE == kMixS32);
};

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

// Two cells per arithmetic instruction for the float32 polynomial: gfx950 issues v_pk_mul_f32 /
// v_pk_add_f32 on float2 at the rate of one scalar op, each lane rounded exactly like the scalar
// instruction.  The faithful float32 mode (numpy's mixed precision) evaluates TPart / SPart /
// al0, p0, lam on float2; everything float64 (p + p0, the division) stays per cell.
typedef float f2 __attribute__((ext_vector_type(2)));
template <>
struct WrightC<f2> : WrightC<float> {};  // scalar constants: they splat in vector arithmetic

template <typename R>
struct Lanes {  // cells per value of arithmetic type R
  static constexpr int n = 1;
  static __device__ __forceinline__ double get(R v, int) { return (double)v; }
  template <typename TIn>
  static __device__ __forceinline__ R make(const TIn* v) { return (R)v[0]; }
};
template <>
struct Lanes<f2> {
  static constexpr int n = 2;
  static __device__ __forceinline__ double get(f2 v, int i) { return (double)(i ? v.y : v.x); }
  template <typename TIn>
  static __device__ __forceinline__ f2 make(const TIn* v) {
    f2 r;
    r.x = (float)v[0];
    r.y = (float)v[1];
    return r;
  }
};

// ---- arithmetic policies -----------------------------------------------------------------------
// ExactOps: numpy's evaluation -- every + - * / is ONE correctly rounded IEEE operation, in the
//   reference's operator order; results are bit-identical to eos/wright.py on the host.
// FusedOps (MLX_FLAG_FMA; momlevel_amd's default for the global sums, opt-in elsewhere): the same
//   expression tree with each "c + a*b" node contracted
//   into one fma and the quotient num/den taken as num*r0*(1 + e + e^2) from the v_rcp_f64 seed r0
//   (e = 1 - den*r0: <= 1 ulp; the Wright denominator lives near 2^19, far from over/underflow).  Not
//   bit-identical to numpy: |rho_fused - rho_numpy| <= a few ulp (parity gate 1e-10 relative).
//   float64 arithmetic; float32 theta/S in numpy's mixed precision keep their float32 polynomial
//   and fuse the float64 tail only (FusedTailOps below).
// ---- the reciprocal ------------------------------------------------------------------------------
// hipcc expands the IEEE f64 division 1.0/den into
//     d' = v_div_scale(den)   y = v_rcp_f64(d')   2 x { e = fma(-d',y,1); y = fma(y,e,y) }
//     n' = v_div_scale(1.0)   q = n'*y   r = fma(-d',q,n')   v_div_fmas(r,y,q)   v_div_fixup
// (11 VALU instructions of a density's ~45).  The scale instructions multiply by a power of two
// ONLY when an operand or the quotient is near the ends of the exponent range; the fix-up only
// patches 0 / inf / NaN operands.  With both scale factors equal to one, q = 1.0*y is y itself
// and the sequence is the seven instructions of rcp_scale_free() below -- bit-identical to the
// IEEE expansion for 2^-1000 < |den| < 2^1000 and trivially equivalent (NaN) for a NaN den, i.e.
// on land.  scripts/check_div.hip measured where it is NOT (profiles/r03_check_div.log): denormal
// den (v_rcp_f64 of a denormal differs from the scaled sequence), |den| within a few binades of
// 2^-1024 and of 2^1022 -- so a class test of the seed alone (is it 0, inf, denormal?) is not
// enough, the guard has to be a two-sided window with a margin.  Two forms:
//   * window on den, two ordered compares (a NaN den passes both): any float64 operands;
//   * ONE v_cmp_class on the seed (0 / inf / denormal?) where the window's margins hold by
//     construction: numpy's float32 polynomial.  There al0, p0, lam are float32 VALUES, so a
//     non-zero den = lam + al0*(p + p0) lies in [2^-462, 2^930] as soon as the level's pressure is 0
//     or 2^-200 <= |p| <= 2^800 (a scalar test per level, p_unsafe()); den == 0 and a float32
//     overflow to inf are what the class test catches.
typedef unsigned long long lanemask_t;  // one bit per lane, in an SGPR pair

// v_cmp_class_f64 mask of what the seed must NOT be: +-inf, +-denormal, +-0
constexpr int kClassNotNanNorNormal = 0x004 | 0x010 | 0x020 | 0x040 | 0x080 | 0x200;

// The guard's state is a LANE MASK in scalar registers: a compare writes its result straight into
// an SGPR pair, the masks of a batch are OR-ed on the scalar unit and the branch is one s_cmp --
// no VALU work beyond the compares themselves.  Written as the instructions they are: through
// __builtin_amdgcn_ballot_w64(...) this compiler (ROCm 7.2 clang) round-trips every lane mask
// through a v_cndmask / v_cmp_ne pair.
__device__ __forceinline__ lanemask_t lanes_not_nan_nor_normal(double y) {
  lanemask_t m;
  asm("v_cmp_class_f64 %0, %1, %2" : "=s"(m) : "v"(y), "s"(kClassNotNanNorNormal));
  return m;
}
// lanes whose |x| is outside (2^-1000, 2^1000); NaN is inside (ordered compares are false on NaN)
__device__ __forceinline__ lanemask_t lanes_outside_window(double x) {
  lanemask_t hi, lo;
  asm("v_cmp_ge_f64 %0, |%1|, %2" : "=s"(hi) : "v"(x), "s"(0x1p+1000));
  asm("v_cmp_le_f64 %0, |%1|, %2" : "=s"(lo) : "v"(x), "s"(0x1p-1000));
  return hi | lo;
}

// -> 1/den by the scale-free sequence (the seed is returned too: the class guard tests it)
__device__ __forceinline__ double rcp_scale_free(double den, double& seed) {
  double y = __builtin_amdgcn_rcp(den);
  seed = y;
  double e = __builtin_fma(-den, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-den, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-den, y, 1.0);  // r = fma(-d', q, n') with q = y, n' = 1
  return __builtin_fma(e, y, y);    // v_div_fmas without the rescale
}

// MLX_TUNE_FMA_ACC (round 4, default on; -DMLX_TUNE_FMA_ACC=0 in A/B builds): the contracting
// policies accumulate c = fma(rho, vol, c) under the "neither is NaN" mask instead of
// term = rho*vol; c += term unless NaN (accumulate<> below): -1 VALU instruction per cell and sum,
// 2-3 % on the instruction-issue-bound kernels (profiles/r04_tune_k1_fused.log).
// (Also measured in round 4 and NOT adopted: one v_rcp_f64 for the three denominators of a cell in
// the one-pass kernels -- Montgomery's trick, 16 reciprocals fewer and 24 multiplications / maxima
// more per 8 cells, 0.4-1.3 % faster, and a denominator from one bad cell or variant would reach
// the others: profiles/r04_batched_reciprocal_negative.txt.)
#ifndef MLX_TUNE_FMA_ACC
#define MLX_TUNE_FMA_ACC 1
#endif
// MLX_TUNE_CUBIC_QUOTIENT (round 4, default on): FusedOps::quotient in five instructions instead
// of six (see there); 3-5 % on the issue-bound kernels (profiles/r04_tune_cubic_quotient.log).
#ifndef MLX_TUNE_CUBIC_QUOTIENT
#define MLX_TUNE_CUBIC_QUOTIENT 1
#endif

struct ExactOps {
  static constexpr bool fused = false;
  static constexpr bool guarded = false;
  static constexpr bool contracts = false;  // may a*b + c be one fma outside the polynomial?
  // (operands may be a float2 vector mixed with scalar constants: the scalars splat)
  template <typename A, typename B, typename C>
  static __device__ __forceinline__ auto mad(A a, B b, C c) -> decltype(a * b + c) {
    return a * b + c;  // two roundings: this file is compiled with contraction off
  }
  // eos/wright.py:47-48: I_denom = 1.0 / den; return (p + p0) * I_denom
  static __device__ __forceinline__ double quotient(double num, double den, lanemask_t&) {
    const double I_denom = 1.0 / den;
    return num * I_denom;
  }
  // lanes that must not take this policy's fast quotient at pressure p (a wave-uniform property)
  static __device__ __forceinline__ lanemask_t p_unsafe(double) { return 0; }
  // the same for a pressure that differs from lane to lane (MLX_P_FULL3D in the fast kernels)
  static __device__ __forceinline__ lanemask_t p_unsafe_lanes(double) { return 0; }
};

// ExactOps with the scale-free reciprocal: the SAME bits as ExactOps while no lane objects.  The
// kernels evaluate a batch of quotients with this policy (quotients<> below), test the objections
// of the whole wave once (a lane mask in SGPRs) and, if any lane objected, redo the batch with
// ExactOps -- so no lane ever keeps a result the guard did not vouch for.
struct ExactFastOps : ExactOps {  // float64 operands: the window on den
  static constexpr bool guarded = true;
  static __device__ __forceinline__ double quotient(double num, double den, lanemask_t& unsafe) {
    unsafe |= lanes_outside_window(den);
    double seed;
    return num * rcp_scale_free(den, seed);
  }
};
struct ExactFastF32Ops : ExactOps {  // float32-valued al0, p0, lam: the class of the seed
  static constexpr bool guarded = true;
  static __device__ __forceinline__ double quotient(double num, double den, lanemask_t& unsafe) {
    double seed;
    const double r = rcp_scale_free(den, seed);
    unsafe |= lanes_not_nan_nor_normal(seed);
    return num * r;
  }
  static __device__ __forceinline__ bool p_ok(double p) {
    const double a = __builtin_fabs(p);
    return !(a > 0x1p+800) && !(a < 0x1p-200 && p != 0.0);  // NaN: every result is NaN anyway
  }
  static __device__ __forceinline__ lanemask_t p_unsafe(double p) {
    return p_ok(p) ? 0 : ~(lanemask_t)0;
  }
  static __device__ __forceinline__ lanemask_t p_unsafe_lanes(double p) {
    return __builtin_amdgcn_ballot_w64(!p_ok(p));  // (once per thread and level, not per cell)
  }
};

struct FusedOps {
  static constexpr bool fused = true;
  static constexpr bool guarded = false;
  static constexpr bool contracts = true;
  template <typename R>
  static __device__ __forceinline__ R mad(R a, R b, R c) {
    return __builtin_fma(a, b, c);
  }
  // Unguarded: nothing here has to match numpy's bits, and the Wright denominator of any ocean
  // state lives near 2^19.  A denominator of exactly 0, inf or a denormal (temperatures and
  // salinities hundreds of units outside the fit's range) yields NaN -- skipped by the sums -- where
  // numpy yields inf / 0.  (Round 3 measured the class guard of ExactFastF32Ops on this policy too:
  // +2 instructions per cell and -4.5 % on the thermosteric sums, profiles/r03_variants_summary.json
  // -- not worth it for inputs that are not sea water.)
  static __device__ __forceinline__ double quotient(double num, double den, lanemask_t&) {
    const double r0 = __builtin_amdgcn_rcp(den);  // 1/den * (1 - e), |e| ~ 2^-23
    if constexpr (MLX_TUNE_CUBIC_QUOTIENT) {
      // num/den = num*r0 * (1 + e + e^2 + ...): the cubic truncation leaves e^3 ~ 2^-69, so
      // q0*(1 + e + e^2) is num/den to <= 1 ulp (the roundings of q0 and of the last fma; checked
      // in exact rational arithmetic over 2e4 Wright-sized operands with seeds as bad as 2^-22:
      // max 0.993 ulp) -- five VALU instructions where Newton + the residual correction took six
      // (round 4).  q0 does not wait for e: the two chains run side by side.
      const double e = __builtin_fma(-den, r0, 1.0);
      const double q0 = num * r0;
      const double t = __builtin_fma(e, e, e);
      return __builtin_fma(q0, t, q0);
    } else {
      const double r = __builtin_fma(__builtin_fma(-den, r0, 1.0), r0, r0);  // ~2^-45
      const double q = num * r;
      return __builtin_fma(__builtin_fma(-den, q, num), r, q);  // exact residual: <= 1/2 ulp
    }
  }
  static __device__ __forceinline__ lanemask_t p_unsafe(double) { return 0; }
  static __device__ __forceinline__ lanemask_t p_unsafe_lanes(double) { return 0; }
};

// MLX_FLAG_FMA on float32 theta/S in numpy's mixed precision (MLX_DTYPE_F32): the POLYNOMIAL stays
// what numpy computes on float32 arrays -- float32, every + and * rounded separately, because that
// rounding (1e-7 relative) is what makes a float32 result a float32 result -- and only the float64
// TAIL is fused: den = fma(al0, p + p0, lam) and FusedOps' quotient.  rho is then within a
// few float64 ulp of numpy's own value on float32 input (FusedOps on upcast values is 1e-7 away),
// for four instructions per cell less than the exact tail.
struct FusedTailOps {
  static constexpr bool fused = false;  // no pressure folding: p enters in float64, as in numpy
  static constexpr bool guarded = false;
  static constexpr bool contracts = true;  // the float64 tail
  template <typename A, typename B, typename C>
  static __device__ __forceinline__ auto mad(A a, B b, C c) -> decltype(a * b + c) {
    return a * b + c;  // the float32 polynomial: two roundings (contraction is off)
  }
  static __device__ __forceinline__ double mad(double a, double b, double c) {
    return __builtin_fma(a, b, c);  // the float64 tail (an exact match beats the template)
  }
  static __device__ __forceinline__ double quotient(double num, double den, lanemask_t& m) {
    return FusedOps::quotient(num, den, m);
  }
  static __device__ __forceinline__ lanemask_t p_unsafe(double) { return 0; }
  static __device__ __forceinline__ lanemask_t p_unsafe_lanes(double) { return 0; }
};

// the policy of MLX_FLAG_FMA for a dtype mode
template <int MODE>
struct FusedFor {
  typedef FusedOps type;
};
template <>
struct FusedFor<kF32Faithful> {
  typedef FusedTailOps type;
};

// the unguarded twin of a policy: what the kernels rerun a wave's work with after an objection
template <typename Ops>
struct SlowOf {
  typedef Ops type;
};
template <>
struct SlowOf<ExactFastOps> {
  typedef ExactOps type;
};
template <>
struct SlowOf<ExactFastF32Ops> {
  typedef ExactOps type;
};

// ---- the Wright polynomial, split into the part that depends on T only, the part that depends
// on S only, and the combination.  This is eos/wright.py:44-46 regrouped along its own
// parentheses:
//     al0 = (A0 + A1*T) + A2*S
//     p0  = (B0 + B4*S) + T*((B1 + T*(B2 + B3*T)) + B5*S)
//     lam = (C0 + C4*S) + T*((C1 + T*(C2 + C3*T)) + C5*S)
// Every sub-expression below is one of the reference's own sub-expressions, rounded the same
// way, so ExactOps stays bit-identical to numpy.  The split is what lets the kernels evaluate a
// part ONCE when its field is held at the reference state (thermosteric: S, halosteric: T) and
// share parts between the three variants in the one-pass decomposition kernel.  FusedOps uses
// the same tree, so all kernels agree bit for bit with each other in that mode too.
template <typename R>
struct TPart {  // functions of T
  R t, a01, tb, tc;
};
template <typename R>
struct SPart {  // functions of S (and, FusedOps only, of the pressure folded into b04)
  R a2s, b04, b5s, c04, c5s;
};

template <typename Ops, typename R>
__device__ __forceinline__ TPart<R> t_part(R T) {
  using K = WrightC<R>;
  TPart<R> h;
  h.t = T;
  h.a01 = Ops::mad(K::A1, T, K::A0);                        // A0 + A1*T
  h.tb = Ops::mad(T, Ops::mad(K::B3, T, K::B2), K::B1);     // B1 + T*(B2 + B3*T)
  h.tc = Ops::mad(T, Ops::mad(K::C3, T, K::C2), K::C1);     // C1 + T*(C2 + C3*T)
  return h;
}

// p_fold: FusedOps folds the pressure into the constant term (B0 + p, rounded once per z level
// or cell) so that p + p0 costs nothing; ExactOps must keep numpy's (p + p0) and passes 0.
template <typename Ops, typename R>
__device__ __forceinline__ SPart<R> s_part(R S, R p_fold) {
  using K = WrightC<R>;
  SPart<R> h;
  h.a2s = K::A2 * S;
  if constexpr (Ops::fused) h.b04 = Ops::mad(K::B4, S, K::B0 + p_fold);
  else h.b04 = Ops::mad(K::B4, S, K::B0);                   // B0 + B4*S
  h.b5s = K::B5 * S;
  h.c04 = Ops::mad(K::C4, S, K::C0);                        // C0 + C4*S
  h.c5s = K::C5 * S;
  return h;
}

// numerator (p + p0) and denominator (lam + al0*(p + p0)) of the density from the two parts,
// eos/wright.py:44-47, for the Lanes<R>::n cells of one arithmetic group
// (p: one pressure per lane of the group -- the level's for every lane, or each cell's own when the
// pressure is a (z,y,x) field)
template <typename Ops, typename R>
__device__ __forceinline__ void wright_numden_lanes(const TPart<R>& a, const SPart<R>& b,
                                                    const double* p, double* num, double* den) {
  const R al0 = a.a01 + b.a2s;
  const R p0 = Ops::mad(a.t, a.tb + b.b5s, b.b04);
  const R lam = Ops::mad(a.t, a.tc + b.c5s, b.c04);
#pragma unroll
  for (int i = 0; i < Lanes<R>::n; ++i) {
    if constexpr (Ops::fused) num[i] = Lanes<R>::get(p0, i);  // p is inside b04
    else num[i] = p[i] + Lanes<R>::get(p0, i);
    den[i] = Ops::mad(Lanes<R>::get(al0, i), num[i], Lanes<R>::get(lam, i));  // lam + al0*(p+p0)
  }
}
template <typename Ops, typename R>
__device__ __forceinline__ void wright_numden_lanes(const TPart<R>& a, const SPart<R>& b, double p,
                                                    double* num, double* den) {
  double pl[Lanes<R>::n];
#pragma unroll
  for (int i = 0; i < Lanes<R>::n; ++i) pl[i] = p;
  wright_numden_lanes<Ops, R>(a, b, pl, num, den);
}

// out[i] = num[i] * (1/den[i]) (eos/wright.py:47-48) for a batch of N cells of one thread.  A
// guarded policy (ExactFastOps, FusedOps) takes the scale-free reciprocal, tests the seeds' class
// flags of the WHOLE WAVE once and, if any lane objected, redoes the batch with the IEEE division
// (SlowOf<Ops>): a wave-uniform scalar branch around eleven instructions per cell that is never
// taken on ocean data.  The volatile asm keeps the compiler from turning the branch into selects
// (it would then evaluate both sides).  Nothing but num/den -- live until the end of the batch
// anyway -- is needed by the fallback, so the guard costs no registers.
template <typename Ops, int N>
__device__ __forceinline__ void quotients(const double* num, const double* den, double* out,
                                          lanemask_t preset = 0) {
  lanemask_t unsafe = preset;  // Ops::p_unsafe(p) of the level: all lanes, or none
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = Ops::quotient(num[i], den[i], unsafe);
  if constexpr (Ops::guarded) {
    if (__builtin_expect(unsafe != 0, 0)) {
      asm volatile("; IEEE-division fallback" ::);
#pragma unroll
      for (int i = 0; i < N; ++i) out[i] = SlowOf<Ops>::type::quotient(num[i], den[i], unsafe);
      asm volatile("; end of fallback" ::);
    }
  }
}

// c += rho*vol unless the product is NaN (derived.py:435-438, skipna).  Policies that contract
// (MLX_FLAG_FMA) take ONE fma under the mask "rho and vol are both numbers" -- two VALU
// instructions (v_cmp_o_f64 rho, vol; v_fma_f64) instead of three (v_mul; v_cmp_o; v_add).  The
// mask differs from "rho*vol is a number" only for inf * 0, which the fused policies do not
// produce on anything resembling sea water (a zero denominator gives NaN there, not inf).
template <typename Ops, bool PREDICATED>
__device__ __forceinline__ void accumulate(double& c, double rho, double vol);

// rho from the two parts: eos/wright.py:44-48.  out[i] = density of lane i (Lanes<R>::n cells)
template <typename Ops, typename R>
__device__ __forceinline__ void wright_combine_lanes(const TPart<R>& a, const SPart<R>& b, double p,
                                                     double* out) {
  double num[Lanes<R>::n], den[Lanes<R>::n];
  wright_numden_lanes<Ops, R>(a, b, p, num, den);
  quotients<Ops, Lanes<R>::n>(num, den, out, Ops::p_unsafe(p));
}

// the guarded exact policy for a dtype mode: see "the reciprocal" above
template <int MODE>
struct ExactFastFor {
  typedef ExactFastOps type;
};
template <>
struct ExactFastFor<kF32Faithful> {
  typedef ExactFastF32Ops type;
};

template <typename Ops, typename R>
__device__ __forceinline__ double wright_combine(const TPart<R>& a, const SPart<R>& b, double p) {
  static_assert(Lanes<R>::n == 1, "scalar form");
  double rho;
  wright_combine_lanes<Ops, R>(a, b, p, &rho);
  return rho;
}

// al0, p0, lam in precision R (double, or float for numpy's float32 inputs: the python-float
// constants are weak scalars and are rounded to float32 first) -- used by the derivatives
template <typename R>
__device__ __forceinline__ void wright_terms(R T, R S, R& al0, R& p0, R& lam) {
  const TPart<R> a = t_part<ExactOps, R>(T);
  const SPart<R> b = s_part<ExactOps, R>(S, R(0));
  al0 = a.a01 + b.a2s;
  p0 = b.b04 + a.t * (a.tb + b.b5s);
  lam = b.c04 + a.t * (a.tc + b.c5s);
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

// arithmetic type of the polynomial when the kernel holds PAIRS of adjacent cells
template <int MODE, int VEC>
struct PolyVec {
  typedef typename PolyType<MODE>::type type;
};
template <>
struct PolyVec<kF32Faithful, 2> {
  typedef f2 type;
};
template <>
struct PolyVec<kF32Faithful, 4> {
  typedef f2 type;
};

// in-situ density, eos/wright.py:44-48.  MODE selects how float32 inputs are treated.
template <int MODE, typename TIn, typename Ops = ExactOps>
__device__ __forceinline__ double wright_density(TIn Tin, TIn Sin, double p) {
  typedef typename PolyType<MODE>::type R;
  static_assert(!(Ops::fused && MODE == kF32Faithful), "FusedOps computes in float64: FusedTailOps");
  const TPart<R> a = t_part<Ops, R>((R)Tin);
  const SPart<R> b = s_part<Ops, R>((R)Sin, Ops::fused ? (R)p : R(0));
  return wright_combine<Ops, R>(a, b, p);
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

// the linear EOS's other functions, eos/linear.py:61-162: the derivatives are the constants
// DRHO_DT = -0.2 and DRHO_DS = 0.8; alpha = -1.0 * (full_like(T, DRHO_DT) / density),
// beta = full_like(T, DRHO_DS) / density -- full_like(T) takes T's dtype, so on float32 input the
// whole quotient is float32 (correctly rounded float32 division: hipcc's default)
template <int MODE, typename TIn>
__device__ __forceinline__ double linear_func(int func, TIn Tin, TIn Sin) {
  if (func == kDrhoDtemp) return -0.2;
  if (func == kDrhoDsal) return 0.8;
  if constexpr (MODE == kF32Faithful) {
    const float rho = 1000.0f + ((-0.2f * Tin) + (0.8f * Sin));
    if (func == kAlpha) return (double)(-1.0f * (-0.2f / rho));
    return (double)(0.8f / rho);
  } else {
    const double rho = linear_density<MODE, TIn>(Tin, Sin);
    if (func == kAlpha) return -1.0 * (-0.2 / rho);
    return 0.8 / rho;
  }
}

// ---- theta and salinity of different dtypes (MODE kMixT32 / kMixS32) ------------------------------
// numpy's promotion on eos/wright.py:44-46 with one float32 and one float64 field: every
// sub-expression that involves only ONE field and constants has that field's dtype -- these are
// exactly the members of TPart / SPart -- and the sums and products that join the two are float64
// (the float32 side widens exactly).  So: each part in its own precision, combine in float64.
// The kernels hand both values over as doubles (the float32 field widened on load, exactly).
template <typename R>
__device__ __forceinline__ TPart<double> widen_part(const TPart<R>& a) {
  return TPart<double>{(double)a.t, (double)a.a01, (double)a.tb, (double)a.tc};
}
template <typename R>
__device__ __forceinline__ SPart<double> widen_part(const SPart<R>& b) {
  return SPart<double>{(double)b.a2s, (double)b.b04, (double)b.b5s, (double)b.c04, (double)b.c5s};
}
// The same rule for the fast kernels, which evaluate the two parts separately (round 4: theta and
// salinity of different dtypes on the 16-byte-load kernels): the float32 field's part in float32
// with numpy's two roundings per multiply-add, widened exactly; the other field's part as the
// kernel's policy evaluates it.  RV is double there (one cell per arithmetic group).
template <int MODE, typename Ops, typename RV>
__device__ __forceinline__ TPart<RV> t_part_m(RV T) {
  if constexpr (MODE == kMixT32) return widen_part(t_part<ExactOps, float>((float)T));
  else return t_part<Ops, RV>(T);
}
template <int MODE, typename Ops, typename RV>
__device__ __forceinline__ SPart<RV> s_part_m(RV S, RV p_fold) {
  if constexpr (MODE == kMixS32) return widen_part(s_part<ExactOps, float>((float)S, 0.0f));
  else return s_part<Ops, RV>(S, p_fold);
}

template <int MODE>
__device__ __forceinline__ double wright_density_mixed(double T, double S, double p) {
  static_assert(IsMixed<MODE>::value, "mixed-dtype modes only");
  typedef typename std::conditional<MODE == kMixT32, float, double>::type RT;
  typedef typename std::conditional<MODE == kMixS32, float, double>::type RS;
  const TPart<double> a = widen_part(t_part<ExactOps, RT>((RT)T));
  const SPart<double> b = widen_part(s_part<ExactOps, RS>((RS)S, RS(0)));
  return wright_combine<ExactOps, double>(a, b, p);
}
// eos/linear.py:55-56: 1000.0 + ((-0.2*T) + (0.8*S)); the products in their field's precision, the
// sum of the two float64, and so is the rest
template <int MODE>
__device__ __forceinline__ double linear_density_mixed(double T, double S) {
  static_assert(IsMixed<MODE>::value, "mixed-dtype modes only");
  typedef typename std::conditional<MODE == kMixT32, float, double>::type RT;
  typedef typename std::conditional<MODE == kMixS32, float, double>::type RS;
  const double dt = (double)((RT)-0.2 * (RT)T), ds = (double)((RS)0.8 * (RS)S);
  return 1000.0 + (dt + ds);
}

// runtime-dispatched EOS function (generic kernels; eos/func are wave-uniform)
// Ops applies to the Wright DENSITY only (the one function on the steric path); the derivatives
// and the linear EOS are always evaluated exactly.
template <int MODE, typename TIn, typename Ops = ExactOps>
__device__ __forceinline__ double eos_eval(int eos, int func, TIn T, TIn S, double p,
                                           double aux = 0.0) {
  if constexpr (IsMixed<MODE>::value) {
    // the steric kernels (K1 / K2) need the density only; the other functions of mixed-dtype
    // operands are mlx_eos_map_promote's (eos_promote.hpp)
    return (eos == kLinear) ? linear_density_mixed<MODE>((double)T, (double)S)
                            : wright_density_mixed<MODE>((double)T, (double)S, p);
  } else {
  if (func == kIbh) {  // pso * (-1.0 / (rho_conv * gravity))
    const double rho = (eos == kLinear) ? linear_density<MODE, TIn>(T, S)
                                        : wright_density<MODE, TIn>(T, S, p);
    return p * (-1.0 / (rho * aux));
  }
  if (eos == kLinear) {
    if (func == kDensity) return linear_density<MODE, TIn>(T, S);
    return linear_func<MODE, TIn>(func, T, S);
  }
  switch (func) {
    case kDensity:
      return wright_density<MODE, TIn, Ops>(T, S, p);
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
  }  // not mixed
}

__device__ __forceinline__ bool is_nan(double x) { return x != x; }

// c += term unless term is NaN (xarray's skipna sums: derived.py:435-438, steric.py:163).  The
// compiler's form of `c += isnan(term) ? 0.0 : term` is v_cmp + 2 x v_cndmask_b32 + v_add_f64 --
// four VALU instructions per cell and sum.  PREDICATED does the same operation in two: the
// compare's lane mask goes into EXEC for the one add (lanes with a NaN term keep their sum
// untouched) and EXEC is put back; the two scalar instructions issue beside other waves' vector
// work.  Identical results bit for bit: the select form adds +0.0 in the skipped lanes, and
// c + 0.0 == c.  Measured in one process per library on one box (profiles/r03_tune_skipna_*.log):
// +2-5 % on the VALU-bound kernels (every float32 kernel, the float64 held-field sums, the one-pass
// kernels), -2 % on the HBM-bound float64 steric sum -- so the kernels pick per instantiation.
// (Not volatile: a pure function of c and term; the compiler stays free to move loads and stores
// across it, which K2's interleaving of delta_rho stores with arithmetic depends on.)
template <bool PREDICATED>
__device__ __forceinline__ void add_skipna(double& c, double term) {
  if constexpr (PREDICATED) {
    lanemask_t saved;
    asm("v_cmp_o_f64 vcc, %2, %2\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_add_f64 %0, %0, %2\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(c), "=&s"(saved)
        : "v"(term)
        : "vcc", "scc");  // (s_and_saveexec also writes SCC: the compiler must not keep a
                          //  compare's result live across this block -- found in round 4 when the
                          //  fma form below moved an s_cmp ... s_cselect pair around such a block)
  } else {
    c += is_nan(term) ? 0.0 : term;
  }
}

// (The fused form tests its OPERANDS, not the product: inf * 0 would slip through -- K1 therefore
//  turns zero volumes into NaN volumes when it loads them, momlevel_hip.hip.)
template <typename Ops, bool PREDICATED>
__device__ __forceinline__ void accumulate(double& c, double rho, double vol) {
  if constexpr (Ops::contracts && MLX_TUNE_FMA_ACC) {
    if constexpr (PREDICATED) {
      lanemask_t saved;
      asm("v_cmp_o_f64 vcc, %2, %3\n\t"
          "s_and_saveexec_b64 %1, vcc\n\t"
          "v_fma_f64 %0, %2, %3, %0\n\t"
          "s_mov_b64 exec, %1"
          : "+v"(c), "=&s"(saved)
          : "v"(rho), "v"(vol)
          : "vcc", "scc");
    } else {
      const double s = __builtin_fma(rho, vol, c);
      c = (rho == rho && vol == vol) ? s : c;
    }
  } else {
    add_skipna<PREDICATED>(c, rho * vol);  // derived.py:435: the product, then the skipna sum
  }
}

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
