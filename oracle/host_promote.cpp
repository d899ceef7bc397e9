// HOST build of mlx_eos_map_promote (include/momlevel_hip.h).  TEST INFRASTRUCTURE ONLY.
//
// numpy's type promotion for the EOS functions is carried by C++ types in
// momlevel_amd/csrc/eos_promote.hpp, a plain C++14 header: this file instantiates the SAME
// restatement for the host (g++, -ffp-contract=off) so that the host build exports the whole ABI.
// It is therefore not an independent restatement of that entry point -- the independent checkers
// of the promote path are numpy itself (oracle/momlevel_numpy.py evaluates any dtype mix natively)
// and the vectors the reference's eos/wright.py / eos/linear.py produced on mixed dtypes
// (tests/golden/wright_vectors.npz, `mix_*`).
#include <stddef.h>
#include <stdint.h>

#include "../include/momlevel_hip.h"
#include "../momlevel_amd/csrc/eos_promote.hpp"

extern "C" int mlxh_fail(int code, const char *msg);

namespace {
using mlx::np::Weak;

struct Operand {
  const void *ptr;
  int64_t stride;
  double weak;
};

template <typename X>
inline X load(const Operand &o, int64_t i) {
  return static_cast<const X *>(o.ptr)[i * o.stride];
}
template <>
inline Weak load<Weak>(const Operand &o, int64_t) {
  return Weak{o.weak};
}

struct Call {
  Operand T, S, p;
  int eos, func;
  double gravity;
  int64_t n;
  void *out;
  bool is_f32;
};

template <typename TT, typename TS, typename TP>
void go(Call &c) {
  bool f32 = false;  // a property of the types and of (eos, func): the same for every cell
  (void)mlx::np::eval<TT, TS, TP>(c.eos, c.func, TT{1}, TS{1}, TP{1}, 1.0, &f32);
  c.is_f32 = f32;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < c.n; ++i) {
    bool unused;
    const double r = mlx::np::eval<TT, TS, TP>(c.eos, c.func, load<TT>(c.T, i), load<TS>(c.S, i),
                                               load<TP>(c.p, i), c.gravity, &unused);
    if (f32) static_cast<float *>(c.out)[i] = (float)r;  // numpy's result dtype; exact
    else static_cast<double *>(c.out)[i] = r;
  }
}
template <typename TT, typename TS>
void by_p(Call &c, int kp) {
  if (kp == MLX_KIND_F64) go<TT, TS, double>(c);
  else if (kp == MLX_KIND_F32) go<TT, TS, float>(c);
  else go<TT, TS, Weak>(c);
}
template <typename TT>
void by_s(Call &c, int ks, int kp) {
  if (ks == MLX_KIND_F64) by_p<TT, double>(c, kp);
  else if (ks == MLX_KIND_F32) by_p<TT, float>(c, kp);
  else by_p<TT, Weak>(c, kp);
}

int operand(const void *ptr, int kind, int64_t stride, Operand *o) {
  if (kind != MLX_KIND_F64 && kind != MLX_KIND_F32 && kind != MLX_KIND_WEAK)
    return mlxh_fail(MLX_E_ENUM, "operand kind must be MLX_KIND_F64, _F32 or _WEAK");
  o->ptr = nullptr;
  o->stride = 0;
  o->weak = 0.0;
  if (!ptr) return mlxh_fail(MLX_E_NULL, "T, S and (unless the EOS is linear) p must not be NULL");
  if (kind == MLX_KIND_WEAK) {
    o->weak = *static_cast<const double *>(ptr);
    return 0;
  }
  if (stride != 0 && stride != 1) return mlxh_fail(MLX_E_SHAPE, "operand stride must be 0 or 1");
  if (reinterpret_cast<uintptr_t>(ptr) % (kind == MLX_KIND_F64 ? 8 : 4))
    return mlxh_fail(MLX_E_ALIGN, "operand not element-aligned");
  o->ptr = ptr;
  o->stride = stride;
  return 0;
}
}  // namespace

extern "C" int mlx_eos_map_promote(const void *T, int kind_T, int64_t stride_T, const void *S,
                                   int kind_S, int64_t stride_S, const void *p, int kind_p,
                                   int64_t stride_p, int eos, int func, double gravity, int64_t n,
                                   void *out, int *out_kind, void *stream) {
  (void)stream;
  if (eos != MLX_EOS_WRIGHT && eos != MLX_EOS_LINEAR) return mlxh_fail(MLX_E_ENUM, "unknown eos");
  if (func < MLX_FUNC_DENSITY || func > MLX_FUNC_DENSITY_REF) return mlxh_fail(MLX_E_ENUM, "unknown func");
  if (func == MLX_FUNC_DENSITY_REF && eos != MLX_EOS_LINEAR)
    return mlxh_fail(MLX_E_ENUM, "MLX_FUNC_DENSITY_REF is eos.linear.density's rho_ref form");
  if (n <= 0) return mlxh_fail(MLX_E_SHAPE, "n must be > 0");
  if (n > ((int64_t)1 << 38)) return mlxh_fail(MLX_E_SHAPE, "n too large");
  if (!out || !out_kind) return mlxh_fail(MLX_E_NULL, "out and out_kind must not be NULL");
  if (reinterpret_cast<uintptr_t>(out) % 8) return mlxh_fail(MLX_E_ALIGN, "out not 8-byte aligned");
  Call c;
  if (int rc = operand(T, kind_T, stride_T, &c.T)) return rc;
  if (int rc = operand(S, kind_S, stride_S, &c.S)) return rc;
  if (eos == MLX_EOS_WRIGHT || func == MLX_FUNC_IBH || func == MLX_FUNC_DENSITY_REF) {
    if (int rc = operand(p, kind_p, stride_p, &c.p)) return rc;
  } else {  // eos/linear.py never reads the pressure
    static const double zero = 0.0;
    kind_p = MLX_KIND_WEAK;
    if (int rc = operand(&zero, kind_p, 0, &c.p)) return rc;
  }
  c.eos = eos;
  c.func = func;
  c.gravity = gravity;
  c.n = n;
  c.out = out;
  c.is_f32 = false;
  if (kind_T == MLX_KIND_F64) by_s<double>(c, kind_S, kind_p);
  else if (kind_T == MLX_KIND_F32) by_s<float>(c, kind_S, kind_p);
  else by_s<Weak>(c, kind_S, kind_p);
  *out_kind = c.is_f32 ? MLX_KIND_F32 : MLX_KIND_F64;
  return 0;
}
