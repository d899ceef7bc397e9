// momlevel_strat.hip -- the stratification diagnostics that consume alpha and beta:
//
//   derived.calc_n2              (src/momlevel/derived.py:328-411)   N^2 = g (alpha dT/dz - beta dS/dz)
//   derived.calc_stability_angle (src/momlevel/derived.py:714-766)   Tu  = degrees(arctan((1+R)/(1-R))),
//                                                                    R   = (beta dS/dz) / (alpha dT/dz)
//   derived.adjust_negative_n2   (src/momlevel/derived.py:30-71)
//   derived.calc_wave_speed      (src/momlevel/derived.py:798-831)
//
// The reference evaluates each as a chain of whole-array numpy passes: alpha (an EOS evaluation:
// density AND its temperature derivative), beta (again), two numpy.gradient calls along z
// (xarray's differentiate(edge_order=2)), four more arithmetic passes -- ~40 temporaries of the
// field's size.  k_stratification does it in ONE pass: a thread owns V = 2 horizontally adjacent
// columns of one time step, walks z with a three-level
// window of theta / S in registers -- numpy.gradient's stencil -- and evaluates alpha, beta at the
// centre level with the operator-for-operator device functions of eos_device.hpp.  16 B read +
// 8 B written per cell at float64 (8 + 8 at float32): the traffic of K0, three times its
// arithmetic.  Every value is the one numpy computes: same operations, same order, same dtypes
// (a float32 field's derivative is rounded to float32 before it meets the float64 alpha), except
// the arctan of the stability angle (numpy's libm vs. the device's: <= 2 ulp, inside 1e-10).
//
// Compile: with momlevel_hip.hip (csrc/build.py), -ffp-contract=off.  Not part of the kernel
// sources whose hash guards the committed counter profiles (those are K0 / K1 / K2).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>
#include <type_traits>

#include "../../include/momlevel_hip.h"
#include "eos_device.hpp"
#include "mlx_internal.hpp"

#pragma clang fp contract(off)

namespace mlx {
namespace {

constexpr int kStratBlock = 256;

struct StratArgs {
  const void* T;
  const void* S;
  const double* p;  // may be null for the linear EOS
  int64_t p_stride_t, p_stride_z, p_stride_cell;
  int eos, func;
  const double* coef;  // (nz, 3): numpy.gradient's a, b, c per level (edge rows: the one-sided ones)
  int uniform;         // evenly spaced levels: interior rows are (f[k+1] - f[k-1]) / two_dx
  double two_dx, gravity;
  int64_t nz, plane;
  double* out;
};

template <typename TIn, int V>
struct Pack {
  TIn v[V];
};

template <typename TIn, int V>
__device__ __forceinline__ Pack<TIn, V> load_pack(const TIn* ptr) {
  Pack<TIn, V> r;
  if constexpr (V * sizeof(TIn) == 16) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const u4 raw = __builtin_nontemporal_load(reinterpret_cast<const u4*>(ptr));
    __builtin_memcpy(r.v, &raw, 16);
  } else if constexpr (V * sizeof(TIn) == 8 && V > 1) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    const u2 raw = __builtin_nontemporal_load(reinterpret_cast<const u2*>(ptr));
    __builtin_memcpy(r.v, &raw, 8);
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = ptr[i];
  }
  return r;
}

// numpy.gradient, one output value.  MODE kF32Faithful: the float32 field meets float64
// coefficients (the products and sums are float64) and the result is STORED as float32 -- the
// derivative of a float32 array is a float32 array -- before alpha / beta (float64) see it.
template <int MODE, typename TIn>
__device__ __forceinline__ double gradient(TIn f0, TIn f1, TIn f2, double a, double b, double c,
                                           bool central, double two_dx) {
  if constexpr (MODE == kF32Faithful) {
    if (central) return (double)(float)((double)(f2 - f0) / two_dx);  // the difference is float32
    return (double)(float)((a * (double)f0 + b * (double)f1) + c * (double)f2);
  } else {
    const double d0 = (double)f0, d1 = (double)f1, d2 = (double)f2;
    if (central) return (d2 - d0) / two_dx;
    return (a * d0 + b * d1) + c * d2;
  }
}

// EOS and FUNC are template arguments: as run-time (wave-uniform) switches they left the hot loop a
// maze of branches around four copies of the arithmetic.
template <typename TIn, int V, int MODE, int EOS, int FUNC>
__global__ __launch_bounds__(kStratBlock) void k_stratification(StratArgs g) {
  const int64_t cell0 = ((int64_t)blockIdx.x * kStratBlock + threadIdx.x) * V;
  if (cell0 >= g.plane) return;
  const int64_t t = blockIdx.y;
  const int64_t nz = g.nz, plane = g.plane;
  const TIn* T = static_cast<const TIn*>(g.T) + t * nz * plane + cell0;
  const TIn* S = static_cast<const TIn*>(g.S) + t * nz * plane + cell0;
  double* out = g.out + t * nz * plane + cell0;
  const double* p = g.p ? g.p + t * g.p_stride_t + cell0 * g.p_stride_cell : nullptr;

  Pack<TIn, V> tw[3], sw[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    tw[k] = load_pack<TIn, V>(T + k * plane);
    sw[k] = load_pack<TIn, V>(S + k * plane);
  }

  // level k from the window: `centre` is the window slot that holds level k itself (a
  // compile-time constant: the window lives in registers and cannot be indexed dynamically)
  auto emit = [&](int64_t k, auto centre_c, bool central) {
    constexpr int centre = decltype(centre_c)::value;
    const double a = g.coef[3 * k], b = g.coef[3 * k + 1], c = g.coef[3 * k + 2];
    double res[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const TIn Tc = tw[centre].v[v], Sc = sw[centre].v[v];
      const double pk = p ? p[k * g.p_stride_z + v * g.p_stride_cell] : 0.0;
      const double alpha = eos_eval<MODE, TIn>(EOS, kAlpha, Tc, Sc, pk);
      const double beta = eos_eval<MODE, TIn>(EOS, kBeta, Tc, Sc, pk);
      const double dtdz =
          gradient<MODE, TIn>(tw[0].v[v], tw[1].v[v], tw[2].v[v], a, b, c, central, g.two_dx);
      const double dsdz =
          gradient<MODE, TIn>(sw[0].v[v], sw[1].v[v], sw[2].v[v], a, b, c, central, g.two_dx);
      if constexpr (FUNC == MLX_STRAT_N2) {
        res[v] = g.gravity * ((alpha * dtdz) - (beta * dsdz));  // derived.py:401
      } else {
        const double r = (beta * dsdz) / (alpha * dtdz);  // derived.py:756
        // np.degrees(np.arctan((1 + R) / (1 - R))): rad2deg multiplies by 180 / pi
        res[v] = atan((1.0 + r) / (1.0 - r)) * (180.0 / 3.14159265358979323846);
      }
    }
    double* o = out + k * plane;
    if constexpr (V == 2) {
      typedef __attribute__((ext_vector_type(2))) double d2;
      __builtin_nontemporal_store(d2{res[0], res[1]}, reinterpret_cast<d2*>(o));
    } else if constexpr (V == 4) {
      typedef __attribute__((ext_vector_type(2))) double d2;
      __builtin_nontemporal_store(d2{res[0], res[1]}, reinterpret_cast<d2*>(o));
      __builtin_nontemporal_store(d2{res[2], res[3]}, reinterpret_cast<d2*>(o + 2));
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) o[v] = res[v];
    }
  };

  const bool central = g.uniform != 0;
  using std::integral_constant;
  emit(0, integral_constant<int, 0>{}, false);    // out[0]  = a f[0] + b f[1] + c f[2]
  emit(1, integral_constant<int, 1>{}, central);  // interior, window (0, 1, 2)
  for (int64_t k = 2; k < nz - 1; ++k) {
    tw[0] = tw[1], tw[1] = tw[2], sw[0] = sw[1], sw[1] = sw[2];
    tw[2] = load_pack<TIn, V>(T + (k + 1) * plane);
    sw[2] = load_pack<TIn, V>(S + (k + 1) * plane);
    emit(k, integral_constant<int, 1>{}, central);
  }
  emit(nz - 1, integral_constant<int, 2>{}, false);  // out[-1] = a f[-3] + b f[-2] + c f[-1]
}

template <typename TIn, int V, int MODE, int EOS>
void launch_eos(const StratArgs& g, dim3 grid, hipStream_t st) {
  if (g.func == MLX_STRAT_N2)
    hipLaunchKernelGGL((k_stratification<TIn, V, MODE, EOS, MLX_STRAT_N2>), grid,
                       dim3(kStratBlock), 0, st, g);
  else
    hipLaunchKernelGGL((k_stratification<TIn, V, MODE, EOS, MLX_STRAT_TURNER>), grid,
                       dim3(kStratBlock), 0, st, g);
}

template <typename TIn, int V, int MODE>
void launch(const StratArgs& g, int64_t nt, hipStream_t st) {
  const int64_t per_block = (int64_t)kStratBlock * V;
  dim3 grid((unsigned)((g.plane + per_block - 1) / per_block), (unsigned)nt);
  if (g.eos == MLX_EOS_LINEAR) {
    // (numpy's float32-throughout linear EOS is refused by the entry point)
    if constexpr (MODE != kF32Faithful) launch_eos<TIn, V, MODE, kLinear>(g, grid, st);
  } else {
    launch_eos<TIn, V, MODE, kWright>(g, grid, st);
  }
}

// adjust_negative_n2 (derived.py:30-71) and calc_wave_speed's column sum (derived.py:822) in one
// walk down each column.  Cells whose index along the array's LEADING dimension is 0 get
// NaN -> 1e-8 BEFORE the forward fill (the reference's `adjusted[0] = adjusted[0].fillna(1e-8)`
// indexes dimension 0, whatever it is): the first `lead0_rows` of the nt rows when a dimension
// other than z leads, the surface level when lead0_rows == 0 (z leads, nt == 1).  adjusted: (nt, nz, plane) or null; speed: (nt, plane) or
// null, = sum_z sqrt(adjusted) * dz (skipna) / pi, dz (nz, plane); for a (z, y, x) field
// (lead0_rows == 0) the reference's final `xr.where(n2[0].isnull(), nan, result)` -- n2[0] is then
// the surface -- is applied here; for a (time, z, y, x) field it is k_speed_where_time0's.
// V adjacent columns per thread (2: one 16-byte access per level), kAdjustDepth levels requested
// before the first is used: the walk down a column is a serial chain only through `carried`, the
// loads are independent of it.
#ifndef MLX_TUNE_ADJUST_DEPTH
#define MLX_TUNE_ADJUST_DEPTH 8
#endif
#ifndef MLX_TUNE_ADJUST_NT_STORE
#define MLX_TUNE_ADJUST_NT_STORE 1
#endif
#ifndef MLX_TUNE_ADJUST_V
#define MLX_TUNE_ADJUST_V 2
#endif
constexpr int kAdjustDepth = MLX_TUNE_ADJUST_DEPTH;

template <int V>
__global__ __launch_bounds__(kStratBlock) void k_adjust_n2(const double* n2, int64_t nz,
                                                           int64_t plane, int64_t lead0_rows,
                                                           const double* dz, double* adjusted,
                                                           double* speed) {
  const int64_t cell = ((int64_t)blockIdx.x * kStratBlock + threadIdx.x) * V;
  if (cell >= plane) return;
  const int64_t t = blockIdx.y;
  const double* col = n2 + t * nz * plane + cell;
  const double nan = __builtin_nan("");
  double carried[V], sum[V], surface[V];
#pragma unroll
  for (int v = 0; v < V; ++v) carried[v] = nan, sum[v] = 0.0, surface[v] = 0.0;
  for (int64_t k0 = 0; k0 < nz; k0 += kAdjustDepth) {
    Pack<double, V> x[kAdjustDepth], w[kAdjustDepth];
#pragma unroll
    for (int j = 0; j < kAdjustDepth; ++j)
      if (k0 + j < nz) {
        x[j] = load_pack<double, V>(col + (k0 + j) * plane);
        if (speed) w[j] = load_pack<double, V>(dz + (k0 + j) * plane + cell);
      }
#pragma unroll
    for (int j = 0; j < kAdjustDepth; ++j) {
      const int64_t k = k0 + j;
      if (k >= nz) break;
      const bool lead0 = lead0_rows ? (t < lead0_rows) : (k == 0);
      double m[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const double xv = x[j].v[v];
        if (k == 0) surface[v] = xv;
        double a = (xv <= 0.0) ? nan : xv;  // xr.where(n2 <= 0.0, nan, n2): NaN compares false, stays
        if (lead0 && a != a) a = 1.0e-8;
        if (a != a) a = carried[v];  // ffill(zcoord)
        carried[v] = a;
        m[v] = (xv != xv) ? nan : a;  // adjusted * mask
        if (speed) {
          const double term = sqrt(m[v]) * w[j].v[v];
          if (term == term) sum[v] += term;  // skipna sum, ascending z as numpy's axis reduce
        }
      }
      if (adjusted) {
        double* o = adjusted + (t * nz + k) * plane + cell;
        if constexpr (V == 2) {
          typedef __attribute__((ext_vector_type(2))) double d2;
#if MLX_TUNE_ADJUST_NT_STORE
          __builtin_nontemporal_store(d2{m[0], m[1]}, reinterpret_cast<d2*>(o));
#else
          *reinterpret_cast<d2*>(o) = d2{m[0], m[1]};
#endif
        } else {
          o[0] = m[0];
        }
      }
    }
  }
  if (speed) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const double c1 = sum[v] / 3.14159265358979323846;
      speed[t * plane + cell + v] = (!lead0_rows && surface[v] != surface[v]) ? nan : c1;
    }
  }
}

// calc_wave_speed on a (time, z, y, x) field, as xarray evaluates derived.py:823: the condition
// n2[0] is the FIRST TIME STEP, dims (z, y, x); the speeds have dims (time, y, x); xr.where
// broadcasts the two by name into (z, y, x, time).  out[k, cell, t] = isnan(n2[0, k, cell]) ? nan :
// speed[t, cell].
__global__ __launch_bounds__(kStratBlock) void k_speed_where_time0(const double* n2_t0,
                                                                   const double* speed,
                                                                   int64_t nt, int64_t nz,
                                                                   int64_t plane, double* out) {
  const int64_t i = (int64_t)blockIdx.x * kStratBlock + threadIdx.x;  // (k, cell, t), t fastest
  if (i >= nz * plane * nt) return;
  const int64_t t = i % nt, kc = i / nt;
  const double c = n2_t0[kc];
  out[i] = (c != c) ? __builtin_nan("") : speed[t * plane + kc % plane];
}

}  // namespace
}  // namespace mlx

extern "C" int mlx_stratification(const void* T, const void* S, int dtype, const double* p,
                                  int64_t p_stride_t, int64_t p_stride_z, int64_t p_stride_cell,
                                  int eos, int func, const double* coef, int uniform,
                                  double two_dx, double gravity, int64_t nt, int64_t nz,
                                  int64_t plane, double* out, void* stream) {
  using namespace mlx;
  if (!T || !S || !coef || !out)
    return detail::fail(MLX_E_NULL, "T, S, coef and out must not be NULL");
  if (eos != MLX_EOS_WRIGHT && eos != MLX_EOS_LINEAR) return detail::fail(MLX_E_ENUM, "unknown eos");
  if (func != MLX_STRAT_N2 && func != MLX_STRAT_TURNER)
    return detail::fail(MLX_E_ENUM, "func must be MLX_STRAT_N2 or MLX_STRAT_TURNER");
  if (dtype != MLX_DTYPE_F64 && dtype != MLX_DTYPE_F32 && dtype != MLX_DTYPE_F32_UPCAST)
    return detail::fail(MLX_E_ENUM, "dtype must be MLX_DTYPE_F64, _F32 or _F32_UPCAST");
  if (eos == MLX_EOS_LINEAR && dtype == MLX_DTYPE_F32)
    return detail::fail(MLX_E_ENUM,
                        "linear EOS on float32 fields is float32 throughout in numpy: not built "
                        "(use MLX_DTYPE_F32_UPCAST or float64 fields)");
  if (!p && eos == MLX_EOS_WRIGHT) return detail::fail(MLX_E_NULL, "p must not be NULL for the Wright EOS");
  if (nt <= 0 || plane <= 0) return detail::fail(MLX_E_SHAPE, "nt and plane must be > 0");
  if (nz < 3)  // numpy.gradient: "at least (edge_order + 1) elements are required"
    return detail::fail(MLX_E_SHAPE, "nz must be >= 3 (second-order edges)");
  if (nz > 65535 || nt > 65535) return detail::fail(MLX_E_SHAPE, "nt and nz must be <= 65535");
  if (plane > ((int64_t)1 << 36)) return detail::fail(MLX_E_SHAPE, "plane too large");
  if ((__int128)nt * nz * plane > ((__int128)1 << 40))
    return detail::fail(MLX_E_SHAPE, "nt*nz*plane too large");
  if (p_stride_t < 0 || p_stride_z < 0 || p_stride_cell < 0 || p_stride_cell > 1)
    return detail::fail(MLX_E_SHAPE, "pressure strides must be >= 0 (cell stride 0 or 1)");
  const bool f64 = dtype == MLX_DTYPE_F64;
  const uintptr_t esz = f64 ? 8 : 4;
  if (reinterpret_cast<uintptr_t>(T) % esz || reinterpret_cast<uintptr_t>(S) % esz)
    return detail::fail(MLX_E_ALIGN, "T/S not element-aligned");
  if (reinterpret_cast<uintptr_t>(out) % 8 || reinterpret_cast<uintptr_t>(coef) % 8 ||
      (p && reinterpret_cast<uintptr_t>(p) % 8))
    return detail::fail(MLX_E_ALIGN, "out / coef / p not 8-byte aligned");
  if (uniform && !(two_dx == two_dx && two_dx != 0.0))
    return detail::fail(MLX_E_SHAPE, "uniform spacing needs a non-zero two_dx");

  StratArgs g{T, S, p, p_stride_t, p_stride_z, p_stride_cell, eos, func, coef, uniform, two_dx,
              gravity, nz, plane, out};
  hipStream_t st = static_cast<hipStream_t>(stream);
#ifndef MLX_TUNE_STRAT_V64
#define MLX_TUNE_STRAT_V64 2
#endif
#ifndef MLX_TUNE_STRAT_V32
#define MLX_TUNE_STRAT_V32 2
#endif
  // cells per thread: two of either type (same-box A/B, scripts/ab_n2.py: four float32 cells per
  // thread -- one 16-byte load -- cost a wave of occupancy at 116 VGPRs and 11 %: 7.93 vs 7.04-7.17 ms;
  // one float64 cell per thread loses 12 %: 8.87 vs 9.94-10.0 ms).  Issue-bound: the load width does
  // not matter, the registers do
  constexpr int V64 = MLX_TUNE_STRAT_V64, V32 = MLX_TUNE_STRAT_V32;
  const int V = f64 ? V64 : V32;
  const bool wide = plane % V == 0 && reinterpret_cast<uintptr_t>(T) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(S) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(out) % 16 == 0;
  if (dtype == MLX_DTYPE_F64) {
    if (wide) launch<double, V64, kF64>(g, nt, st);
    else launch<double, 1, kF64>(g, nt, st);
  } else if (dtype == MLX_DTYPE_F32) {
    if (wide) launch<float, V32, kF32Faithful>(g, nt, st);
    else launch<float, 1, kF32Faithful>(g, nt, st);
  } else {
    if (wide) launch<float, V32, kF32Upcast>(g, nt, st);
    else launch<float, 1, kF32Upcast>(g, nt, st);
  }
  return detail::hip_status(hipGetLastError(), "mlx_stratification launch");
}

extern "C" int mlx_adjust_negative_n2(const double* n2, int64_t nt, int64_t nz, int64_t plane,
                                      int64_t lead0_rows, const double* dz, double* adjusted,
                                      double* speed, void* stream) {
  using namespace mlx;
  if (!n2) return detail::fail(MLX_E_NULL, "n2 must not be NULL");
  if (!adjusted && !speed) return detail::fail(MLX_E_NULL, "one of adjusted / speed is required");
  if (speed && !dz) return detail::fail(MLX_E_NULL, "speed needs dz");
  if (nt <= 0 || nz <= 0 || plane <= 0) return detail::fail(MLX_E_SHAPE, "nt, nz, plane must be > 0");
  if (nt > 65535 || nz > 65535) return detail::fail(MLX_E_SHAPE, "nt and nz must be <= 65535");
  if (lead0_rows < 0 || lead0_rows > nt || (lead0_rows == 0 && nt != 1))
    return detail::fail(MLX_E_SHAPE, "lead0_rows must be in 1..nt, or 0 with nt == 1");
  if (plane > ((int64_t)1 << 36) || (__int128)nt * nz * plane > ((__int128)1 << 40))
    return detail::fail(MLX_E_SHAPE, "nt*nz*plane too large");
  for (const void* q : {(const void*)n2, (const void*)dz, (const void*)adjusted, (const void*)speed})
    if (q && reinterpret_cast<uintptr_t>(q) % 8) return detail::fail(MLX_E_ALIGN, "pointer not 8-byte aligned");
  bool wide = plane % 2 == 0 && MLX_TUNE_ADJUST_V == 2;
  for (const void* q : {(const void*)n2, (const void*)dz, (const void*)adjusted})
    if (q && reinterpret_cast<uintptr_t>(q) % 16) wide = false;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (wide) {
    dim3 grid((unsigned)((plane / 2 + kStratBlock - 1) / kStratBlock), (unsigned)nt);
    hipLaunchKernelGGL(k_adjust_n2<2>, grid, dim3(kStratBlock), 0, st, n2, nz, plane, lead0_rows, dz,
                       adjusted, speed);
  } else {
    dim3 grid((unsigned)((plane + kStratBlock - 1) / kStratBlock), (unsigned)nt);
    hipLaunchKernelGGL(k_adjust_n2<1>, grid, dim3(kStratBlock), 0, st, n2, nz, plane, lead0_rows, dz,
                       adjusted, speed);
  }
  return detail::hip_status(hipGetLastError(), "mlx_adjust_negative_n2 launch");
}

extern "C" int mlx_wave_speed_where_time0(const double* n2_t0, const double* speed, int64_t nt,
                                          int64_t nz, int64_t plane, double* out, void* stream) {
  using namespace mlx;
  if (!n2_t0 || !speed || !out) return detail::fail(MLX_E_NULL, "n2_t0, speed and out must not be NULL");
  if (nt <= 0 || nz <= 0 || plane <= 0) return detail::fail(MLX_E_SHAPE, "nt, nz, plane must be > 0");
  // (each factor bounded first: the product of three arbitrary int64 overflows even __int128)
  if (nt > 65535 || nz > 65535 || plane > ((int64_t)1 << 36) ||
      (__int128)nt * nz * plane > ((__int128)1 << 38))
    return detail::fail(MLX_E_SHAPE, "nt*nz*plane too large");
  for (const void* q : {(const void*)n2_t0, (const void*)speed, (const void*)out})
    if (reinterpret_cast<uintptr_t>(q) % 8) return detail::fail(MLX_E_ALIGN, "pointer not 8-byte aligned");
  const int64_t n = nt * nz * plane;
  hipLaunchKernelGGL(k_speed_where_time0, dim3((unsigned)((n + kStratBlock - 1) / kStratBlock)),
                     dim3(kStratBlock), 0, static_cast<hipStream_t>(stream), n2_t0, speed, nt, nz,
                     plane, out);
  return detail::hip_status(hipGetLastError(), "mlx_wave_speed_where_time0 launch");
}
