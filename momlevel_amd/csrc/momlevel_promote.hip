// momlevel_promote.hip -- the pointwise EOS functions for EVERY dtype combination numpy accepts.
//
// mlx_eos_map (momlevel_hip.hip) covers the two combinations the steric path streams: float64
// everything, and float32 theta/S with a float64 pressure.  The reference's EOS functions are
// numpy expressions, so they also take -- and calc_pdens / inverse_barometer / direct calls DO
// hand them -- a python-float pressure (the whole expression then stays float32 on float32
// fields), float32 pressure arrays, theta and salinity of different dtypes, python floats for
// theta or salinity.  k_eos_promote evaluates eos_promote.hpp's operator-for-operator restatement
// (numpy's promotion rules carried by the C++ types) for all 27 combinations: one thread per
// cell, coalesced loads, pointwise, no reuse -- bandwidth-bound like K0, and not on the steric
// path.
//
// Compile: with momlevel_hip.hip (csrc/build.py), -ffp-contract=off.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <initializer_list>
#include <type_traits>

#include "../../include/momlevel_hip.h"
#include "eos_promote.hpp"
#include "mlx_internal.hpp"

#pragma clang fp contract(off)

namespace mlx {

static_assert(np::kEosWright == MLX_EOS_WRIGHT && np::kEosLinear == MLX_EOS_LINEAR, "eos enum");
static_assert(np::kFnDensity == MLX_FUNC_DENSITY && np::kFnDrhoDtemp == MLX_FUNC_DRHO_DTEMP &&
                  np::kFnDrhoDsal == MLX_FUNC_DRHO_DSAL && np::kFnAlpha == MLX_FUNC_ALPHA &&
                  np::kFnBeta == MLX_FUNC_BETA && np::kFnIbh == MLX_FUNC_IBH &&
                      np::kFnDensityRef == MLX_FUNC_DENSITY_REF,
              "func enum");

constexpr int kPromoteBlock = 256;

struct PromoteOperand {
  const void* ptr;  // device array (F32 / F64), unused for a weak scalar
  int64_t stride;   // 1, or 0 = one value for every cell
  double weak;      // the python float
};

template <typename X>
__device__ __forceinline__ X promote_load(const PromoteOperand& o, int64_t i) {
  return static_cast<const X*>(o.ptr)[i * o.stride];
}
template <>
__device__ __forceinline__ np::Weak promote_load<np::Weak>(const PromoteOperand& o, int64_t) {
  return np::Weak{o.weak};
}

// The result is stored in numpy's result dtype: `out` is n float32 values when the expression is
// float32 (wave-uniform: a property of the types and of (eos, func)), n float64 values otherwise.
//
// VEC cells per thread, chosen by the host so that the WIDEST array operand (or the result) of a
// thread is one 16-byte access: 4 cells when everything is float32, 2 as soon as a float64 array
// takes part (its floats then come as 8-byte loads) -- a wave moves 1 KiB per widest load
// instruction instead of 256 / 512 B, which is what a pointwise map's bandwidth depends on
// (cdna_hip_programming.md G2; measured: 0.51 -> 0.75 of the HBM peak on calc_pdens' combination).
// Needs every array operand and the result 16-byte aligned; unaligned views take VEC = 1, and the
// last, partial group of a launch goes cell by cell.
template <typename X, int VEC>
struct PromoteVals {
  X v[VEC];
};

typedef float promote_f4 __attribute__((ext_vector_type(4)));

template <typename X, int VEC>
__device__ __forceinline__ PromoteVals<X, VEC> promote_load_group(const PromoteOperand& o, int64_t i0) {
  PromoteVals<X, VEC> r;
  if constexpr (std::is_same<X, np::Weak>::value) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) r.v[k] = np::Weak{o.weak};
  } else {
    const X* base = static_cast<const X*>(o.ptr);
    if (o.stride == 0) {
      const X x = base[0];
#pragma unroll
      for (int k = 0; k < VEC; ++k) r.v[k] = x;
    } else if constexpr (sizeof(X) * VEC == 16) {  // 4 floats / 2 doubles: one 16-byte load
      const promote_f4 raw = __builtin_nontemporal_load(reinterpret_cast<const promote_f4*>(base + i0));
      __builtin_memcpy(&r, &raw, 16);
    } else if constexpr (sizeof(X) * VEC == 8) {   // 2 floats beside doubles: one 8-byte load
      const double raw = __builtin_nontemporal_load(reinterpret_cast<const double*>(base + i0));
      __builtin_memcpy(&r, &raw, 8);
    } else {
      static_assert(VEC == 1, "groups are 8 or 16 bytes per operand");
      r.v[0] = base[i0];
    }
  }
  return r;
}

template <typename TT, typename TS, typename TP, int VEC>
__global__ __launch_bounds__(kPromoteBlock) void k_eos_promote(PromoteOperand T, PromoteOperand S,
                                                               PromoteOperand p, int eos, int func,
                                                               double gravity, int64_t n,
                                                               void* __restrict__ out) {
  const int64_t i0 = ((int64_t)blockIdx.x * kPromoteBlock + threadIdx.x) * VEC;
  if (i0 >= n) return;
  bool is_f32 = false;
  if (i0 + VEC <= n) {
    const PromoteVals<TT, VEC> a = promote_load_group<TT, VEC>(T, i0);
    const PromoteVals<TS, VEC> b = promote_load_group<TS, VEC>(S, i0);
    const PromoteVals<TP, VEC> c = promote_load_group<TP, VEC>(p, i0);
    double r[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      r[k] = np::eval<TT, TS, TP>(eos, func, a.v[k], b.v[k], c.v[k], gravity, &is_f32);
    if (is_f32) {
      PromoteVals<float, VEC> f;
#pragma unroll
      for (int k = 0; k < VEC; ++k) f.v[k] = (float)r[k];  // exact: r holds float32 values
      float* dst = static_cast<float*>(out) + i0;
      if constexpr (VEC == 4) {
        promote_f4 raw;
        __builtin_memcpy(&raw, &f, 16);
        __builtin_nontemporal_store(raw, reinterpret_cast<promote_f4*>(dst));
      } else if constexpr (VEC == 2) {
        double raw;
        __builtin_memcpy(&raw, &f, 8);
        __builtin_nontemporal_store(raw, reinterpret_cast<double*>(dst));
      } else {
        dst[0] = f.v[0];
      }
    } else {
      double* dst = static_cast<double*>(out) + i0;
      if constexpr (VEC == 1) {
        dst[0] = r[0];
      } else {
#pragma unroll
        for (int q = 0; q < VEC / 2; ++q) {
          promote_f4 raw;
          __builtin_memcpy(&raw, &r[2 * q], 16);
          __builtin_nontemporal_store(raw, reinterpret_cast<promote_f4*>(dst) + q);
        }
      }
    }
    return;
  }
  for (int64_t i = i0; i < n; ++i) {  // the launch's last, partial group
    const double r = np::eval<TT, TS, TP>(eos, func, promote_load<TT>(T, i), promote_load<TS>(S, i),
                                          promote_load<TP>(p, i), gravity, &is_f32);
    if (is_f32) static_cast<float*>(out)[i] = (float)r;
    else static_cast<double*>(out)[i] = r;
  }
}

namespace {

// numpy's result dtype for the combination: evaluated on the host from the same expression types
template <typename TT, typename TS, typename TP>
bool result_is_f32(int eos, int func) {
  bool is_f32 = false;
  (void)np::eval<TT, TS, TP>(eos, func, TT{1}, TS{1}, TP{1}, 1.0, &is_f32);
  return is_f32;
}

struct PromoteCall {
  PromoteOperand T, S, p;
  int eos, func;
  double gravity;
  int64_t n;
  void* out;
  hipStream_t st;
  bool is_f32;
  bool aligned16;  // every array operand and the result 16-byte aligned: several cells per thread
};

template <typename TT, typename TS, typename TP, int VEC>
void promote_launch(PromoteCall& c) {
  const int64_t groups = (c.n + VEC - 1) / VEC, blocks = (groups + kPromoteBlock - 1) / kPromoteBlock;
  hipLaunchKernelGGL((k_eos_promote<TT, TS, TP, VEC>), dim3((unsigned)blocks), dim3(kPromoteBlock), 0,
                     c.st, c.T, c.S, c.p, c.eos, c.func, c.gravity, c.n, c.out);
}

template <typename TT, typename TS, typename TP>
void promote_go(PromoteCall& c) {
  c.is_f32 = result_is_f32<TT, TS, TP>(c.eos, c.func);
  // a float64 array among the operands (the result is then float64 too): 2 cells = 16 bytes of it
  constexpr bool ANY_F64 = std::is_same<TT, double>::value || std::is_same<TS, double>::value ||
                           std::is_same<TP, double>::value;
  if (!c.aligned16) {
    promote_launch<TT, TS, TP, 1>(c);
  } else if constexpr (ANY_F64) {
    promote_launch<TT, TS, TP, 2>(c);
  } else {  // float32 arrays and python floats only
    if (c.is_f32) promote_launch<TT, TS, TP, 4>(c);
    else promote_launch<TT, TS, TP, 2>(c);  // (a float64 result of float32 operands: full_like(weak T))
  }
}

template <typename TT, typename TS>
void promote_p(PromoteCall& c, int kind_p) {
  if (kind_p == MLX_KIND_F64) promote_go<TT, TS, double>(c);
  else if (kind_p == MLX_KIND_F32) promote_go<TT, TS, float>(c);
  else promote_go<TT, TS, np::Weak>(c);
}

template <typename TT>
void promote_s(PromoteCall& c, int kind_S, int kind_p) {
  if (kind_S == MLX_KIND_F64) promote_p<TT, double>(c, kind_p);
  else if (kind_S == MLX_KIND_F32) promote_p<TT, float>(c, kind_p);
  else promote_p<TT, np::Weak>(c, kind_p);
}

// validates one operand and fills its launch record; `what` names it in the error text
int promote_operand(const void* ptr, int kind, int64_t stride, const char* what,
                    PromoteOperand* o) {
  static thread_local char msg[96];
  if (kind != MLX_KIND_F64 && kind != MLX_KIND_F32 && kind != MLX_KIND_WEAK) {
    snprintf(msg, sizeof(msg), "kind of %s must be MLX_KIND_F64, _F32 or _WEAK", what);
    return detail::fail(MLX_E_ENUM, msg);
  }
  o->ptr = nullptr;
  o->stride = 0;
  o->weak = 0.0;
  if (!ptr) {
    snprintf(msg, sizeof(msg), "%s must not be NULL", what);
    return detail::fail(MLX_E_NULL, msg);
  }
  if (kind == MLX_KIND_WEAK) {
    o->weak = *static_cast<const double*>(ptr);  // a HOST pointer, read now
    return 0;
  }
  if (stride != 0 && stride != 1) {
    snprintf(msg, sizeof(msg), "stride of %s must be 0 (one value) or 1", what);
    return detail::fail(MLX_E_SHAPE, msg);
  }
  const size_t es = (kind == MLX_KIND_F64) ? 8 : 4;
  if (reinterpret_cast<uintptr_t>(ptr) % es) {
    snprintf(msg, sizeof(msg), "%s not element-aligned", what);
    return detail::fail(MLX_E_ALIGN, msg);
  }
  o->ptr = ptr;
  o->stride = stride;
  return 0;
}

}  // namespace
}  // namespace mlx

extern "C" int mlx_eos_map_promote(const void* T, int kind_T, int64_t stride_T, const void* S,
                                   int kind_S, int64_t stride_S, const void* p, int kind_p,
                                   int64_t stride_p, int eos, int func, double gravity, int64_t n,
                                   void* out, int* out_kind, void* stream) {
  using namespace mlx;
  if (eos != MLX_EOS_WRIGHT && eos != MLX_EOS_LINEAR) return detail::fail(MLX_E_ENUM, "unknown eos");
  if (func < MLX_FUNC_DENSITY || func > MLX_FUNC_DENSITY_REF)
    return detail::fail(MLX_E_ENUM, "unknown func");
  if (func == MLX_FUNC_DENSITY_REF && eos != MLX_EOS_LINEAR)
    return detail::fail(MLX_E_ENUM, "MLX_FUNC_DENSITY_REF is eos.linear.density's rho_ref form");
  if (n <= 0) return detail::fail(MLX_E_SHAPE, "n must be > 0");
  if (n > ((int64_t)1 << 38)) return detail::fail(MLX_E_SHAPE, "n too large");
  if (!out || !out_kind) return detail::fail(MLX_E_NULL, "out and out_kind must not be NULL");
  if (reinterpret_cast<uintptr_t>(out) % 8) return detail::fail(MLX_E_ALIGN, "out not 8-byte aligned");
  PromoteCall c;
  const bool p_read = (eos == MLX_EOS_WRIGHT) || func == MLX_FUNC_IBH || func == MLX_FUNC_DENSITY_REF;
  if (int rc = promote_operand(T, kind_T, stride_T, "T", &c.T)) return rc;
  if (int rc = promote_operand(S, kind_S, stride_S, "S", &c.S)) return rc;
  if (p_read) {
    if (int rc = promote_operand(p, kind_p, stride_p, "p", &c.p)) return rc;
  } else {
    // eos/linear.py never reads the pressure: whatever was passed (NULL included) is not touched and
    // takes no part in the promotion
    static const double zero = 0.0;
    kind_p = MLX_KIND_WEAK;
    if (int rc = promote_operand(&zero, kind_p, 0, "p", &c.p)) return rc;
  }
  c.eos = eos;
  c.func = func;
  c.gravity = gravity;
  c.n = n;
  c.out = out;
  c.st = static_cast<hipStream_t>(stream);
  c.is_f32 = false;
  c.aligned16 = (reinterpret_cast<uintptr_t>(out) % 16) == 0;
  for (const PromoteOperand* o : {&c.T, &c.S, &c.p})
    if (o->ptr && o->stride == 1 && reinterpret_cast<uintptr_t>(o->ptr) % 16) c.aligned16 = false;
  if (kind_T == MLX_KIND_F64) promote_s<double>(c, kind_S, kind_p);
  else if (kind_T == MLX_KIND_F32) promote_s<float>(c, kind_S, kind_p);
  else promote_s<np::Weak>(c, kind_S, kind_p);
  *out_kind = c.is_f32 ? MLX_KIND_F32 : MLX_KIND_F64;
  return detail::hip_status(hipGetLastError(), "mlx_eos_map_promote launch");
}
