/*
 * host_abi.c -- HOST (CPU) build of the C ABI declared in include/momlevel_hip.h.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product never links or loads it.
 *
 * SURVEY.md 8b asks for "a host (CPU) build of the same ABI ... used for tests in GPU-less
 * containers and as the timed CPU restatement".  This file is that build: every symbol of the
 * header, same argument lists, same status codes -- but all pointers are HOST pointers, `stream`
 * and the workspace are ignored, and the arithmetic is plain C restated from the reference:
 *   eos/wright.py:6-20,44-48,74-83,108-117,142,165   Wright EOS and its derivatives
 *   eos/linear.py:55-56                              linear EOS density
 *   derived.py:295-323 (calc_dz), 435-438 (calc_masso), 789 (calc_volo)
 *   steric.py:115-125 (held fields), 134-147 (global), 150-166 (local)
 *   util.py:85-92 (annual weighted mean), dynamic.py:34-36 (inverse barometer)
 * Compiled with -ffp-contract=off: every + - * / is one rounding, as numpy evaluates the
 * reference, so pointwise outputs are bit-identical to oracle/momlevel_numpy.py (and to the HIP
 * kernels); sums are added in plain ascending order (z, then plane), within 1e-12 of numpy's
 * pairwise sums.  It is a SECOND, independent restatement: the tests check it against the numpy
 * oracle and the reference's goldens on CPU, and against the HIP library on the GPU.
 * MLX_FLAG_FMA is not offered here (exact arithmetic only); MLX_FLAG_SKIP_DRY is accepted and
 * changes nothing, as on the device.
 *
 * Build: make -C oracle libmomlevel_host.so   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../include/momlevel_hip.h"

static _Thread_local char g_err[512] = "";

static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

/* host_promote.cpp (mlx_eos_map_promote) reports through the same buffer */
int mlxh_fail(int code, const char *msg) { return fail(code, msg); }

/* ---- eos/wright.py:6-20 ------------------------------------------------------------------- */
#define A0 7.057924e-4
#define A1 3.480336e-7
#define A2 -1.112733e-7
#define B0 5.790749e8
#define B1 3.516535e6
#define B2 -4.002714e4
#define B3 2.084372e2
#define B4 5.944068e5
#define B5 -9.643486e3
#define C0 1.704853e5
#define C1 7.904722e2
#define C2 -7.984422
#define C3 5.140652e-2
#define C4 -2.302158e2
#define C5 -3.079464

/* al0, p0, lam in float64 (eos/wright.py:44-46) */
static inline void terms64(double T, double S, double *al0, double *p0, double *lam) {
  *al0 = (A0 + A1 * T) + A2 * S;
  *p0 = (B0 + B4 * S) + T * ((B1 + T * (B2 + B3 * T)) + B5 * S);
  *lam = (C0 + C4 * S) + T * ((C1 + T * (C2 + C3 * T)) + C5 * S);
}
/* the same in float32: numpy keeps float32 arrays float32 against python-float constants */
static inline void terms32(float T, float S, float *al0, float *p0, float *lam) {
  *al0 = ((float)A0 + (float)A1 * T) + (float)A2 * S;
  *p0 = ((float)B0 + (float)B4 * S) + T * (((float)B1 + T * ((float)B2 + (float)B3 * T)) + (float)B5 * S);
  *lam = ((float)C0 + (float)C4 * S) + T * (((float)C1 + T * ((float)C2 + (float)C3 * T)) + (float)C5 * S);
}
static inline double rho_from_terms(double al0, double p0, double lam, double p) {
  const double pp0 = p + p0;
  const double I_denom = 1.0 / (lam + al0 * pp0);
  return pp0 * I_denom;
}

/* one theta/S value of either storage type */
typedef struct { double d; float f; int is_f32_faithful; } Val;
/* dtype of ONE field (theta: is_T) for a MLX_DTYPE_* code; the mixed codes give each field its own */
static inline int field_dtype(int dtype, int is_T) {
  if (dtype == MLX_DTYPE_T32_S64) return is_T ? MLX_DTYPE_F32 : MLX_DTYPE_F64;
  if (dtype == MLX_DTYPE_T64_S32) return is_T ? MLX_DTYPE_F64 : MLX_DTYPE_F32;
  return dtype;
}
static inline Val load(const void *base, int64_t i, int dtype) {
  Val v;
  if (dtype == MLX_DTYPE_F64) { v.d = ((const double *)base)[i]; v.f = 0.0f; v.is_f32_faithful = 0; }
  else { v.f = ((const float *)base)[i]; v.d = (double)v.f; v.is_f32_faithful = (dtype == MLX_DTYPE_F32); }
  return v;
}

/* theta and salinity of different dtypes (MLX_DTYPE_T32_S64 / _T64_S32): numpy keeps every
 * sub-expression of eos/wright.py:44-46 that involves ONE field in that field's dtype and joins the
 * two in float64 */
static inline double density_mixed(Val T, Val S, double p) {
  double a01, tb, tc, a2s, b04, b5s, c04, c5s;
  if (T.is_f32_faithful) {
    a01 = (double)((float)A0 + (float)A1 * T.f);
    tb = (double)((float)B1 + T.f * ((float)B2 + (float)B3 * T.f));
    tc = (double)((float)C1 + T.f * ((float)C2 + (float)C3 * T.f));
  } else {
    a01 = A0 + A1 * T.d;
    tb = B1 + T.d * (B2 + B3 * T.d);
    tc = C1 + T.d * (C2 + C3 * T.d);
  }
  if (S.is_f32_faithful) {
    a2s = (double)((float)A2 * S.f);
    b04 = (double)((float)B0 + (float)B4 * S.f);
    b5s = (double)((float)B5 * S.f);
    c04 = (double)((float)C0 + (float)C4 * S.f);
    c5s = (double)((float)C5 * S.f);
  } else {
    a2s = A2 * S.d;
    b04 = B0 + B4 * S.d;
    b5s = B5 * S.d;
    c04 = C0 + C4 * S.d;
    c5s = C5 * S.d;
  }
  const double al0 = a01 + a2s, p0 = b04 + T.d * (tb + b5s), lam = c04 + T.d * (tc + c5s);
  return rho_from_terms(al0, p0, lam, p);
}

static inline double density(Val T, Val S, double p) {
  if (T.is_f32_faithful != S.is_f32_faithful) return density_mixed(T, S, p);
  if (T.is_f32_faithful) {
    float a, b, c;
    terms32(T.f, S.f, &a, &b, &c);
    return rho_from_terms((double)a, (double)b, (double)c, p);
  }
  double a, b, c;
  terms64(T.d, S.d, &a, &b, &c);
  return rho_from_terms(a, b, c, p);
}
static inline double drho_dtemp(Val T, Val S, double p) { /* eos/wright.py:74-83 */
  if (T.is_f32_faithful) {
    float al0, p0, lam;
    terms32(T.f, S.f, &al0, &p0, &lam);
    const double pp0 = p + (double)p0;
    double I2 = 1.0 / ((double)lam + (double)al0 * pp0);
    I2 = I2 * I2;
    const float two_b2 = (float)(2.0 * B2), three_b3 = (float)(3.0 * B3);
    const float two_c2 = (float)(C2 * 2.0), three_c3 = (float)(C3 * 3.0);
    const float a = lam * (((float)B1 + T.f * (two_b2 + three_b3 * T.f)) + (float)B5 * S.f);
    const float cp = ((float)C1 + T.f * (two_c2 + three_c3 * T.f)) + (float)C5 * S.f;
    const double b = pp0 * (pp0 * A1 + (double)cp);
    return I2 * ((double)a - b);
  }
  double al0, p0, lam;
  terms64(T.d, S.d, &al0, &p0, &lam);
  const double pp0 = p + p0;
  double I2 = 1.0 / (lam + al0 * pp0);
  I2 = I2 * I2;
  const double a = lam * ((B1 + T.d * (2.0 * B2 + (3.0 * B3) * T.d)) + B5 * S.d);
  const double b = pp0 * (pp0 * A1 + ((C1 + T.d * (C2 * 2.0 + (C3 * 3.0) * T.d)) + C5 * S.d));
  return I2 * (a - b);
}
static inline double drho_dsal(Val T, Val S, double p) { /* eos/wright.py:108-117 */
  if (T.is_f32_faithful) {
    float al0, p0, lam;
    terms32(T.f, S.f, &al0, &p0, &lam);
    const double pp0 = p + (double)p0;
    double I2 = 1.0 / ((double)lam + (double)al0 * pp0);
    I2 = I2 * I2;
    const float a = lam * ((float)B4 + (float)B5 * T.f);
    const float c = (float)C4 + (float)C5 * T.f;
    return I2 * ((double)a - pp0 * (pp0 * A2 + (double)c));
  }
  double al0, p0, lam;
  terms64(T.d, S.d, &al0, &p0, &lam);
  const double pp0 = p + p0;
  double I2 = 1.0 / (lam + al0 * pp0);
  I2 = I2 * I2;
  return I2 * (lam * (B4 + B5 * T.d) - pp0 * (pp0 * A2 + (C4 + C5 * T.d)));
}
static inline double linear_density(Val T, Val S) { /* eos/linear.py:55-56 */
  if (T.is_f32_faithful != S.is_f32_faithful) { /* products in their field's dtype, the rest float64 */
    const double dt = T.is_f32_faithful ? (double)(-0.2f * T.f) : -0.2 * T.d;
    const double ds = S.is_f32_faithful ? (double)(0.8f * S.f) : 0.8 * S.d;
    return 1000.0 + (dt + ds);
  }
  if (T.is_f32_faithful) return (double)(1000.0f + ((-0.2f * T.f) + (0.8f * S.f)));
  return 1000.0 + ((-0.2 * T.d) + (0.8 * S.d));
}
static inline double eos_eval(int eos, int func, Val T, Val S, double p, double aux) {
  if (func == MLX_FUNC_IBH) {
    const double rho = (eos == MLX_EOS_LINEAR) ? linear_density(T, S) : density(T, S, p);
    return p * (-1.0 / (rho * aux));
  }
  if (eos == MLX_EOS_LINEAR) { /* eos/linear.py:61-162 */
    if (func == MLX_FUNC_DENSITY) return linear_density(T, S);
    if (func == MLX_FUNC_DRHO_DTEMP) return -0.2;
    if (func == MLX_FUNC_DRHO_DSAL) return 0.8;
    if (T.is_f32_faithful) { /* full_like(T) is float32: the whole quotient stays float32 */
      const float rho = 1000.0f + ((-0.2f * T.f) + (0.8f * S.f));
      return (func == MLX_FUNC_ALPHA) ? (double)(-1.0f * (-0.2f / rho)) : (double)(0.8f / rho);
    }
    const double rho = linear_density(T, S);
    return (func == MLX_FUNC_ALPHA) ? -1.0 * (-0.2 / rho) : 0.8 / rho;
  }
  switch (func) {
    case MLX_FUNC_DENSITY: return density(T, S, p);
    case MLX_FUNC_DRHO_DTEMP: return drho_dtemp(T, S, p);
    case MLX_FUNC_DRHO_DSAL: return drho_dsal(T, S, p);
    case MLX_FUNC_ALPHA: return -1.0 * (drho_dtemp(T, S, p) / density(T, S, p));
    default: return drho_dsal(T, S, p) / density(T, S, p);
  }
}
static inline double pressure(const double *p, int p_mode, int64_t t, int64_t z, int64_t i,
                              int64_t nz, int64_t plane) {
  switch (p_mode) {
    case MLX_P_SCALAR: return p[0];
    case MLX_P_ZPROF: return p[z];
    case MLX_P_FULL3D: return p[z * plane + i];
    default: return p[(t * nz + z) * plane + i];
  }
}
static inline double nan0(double x) { return (x == x) ? x : 0.0; }
static inline double canonical_nan(void) { return NAN; }

static int check_common_m(const void *T, const void *S, int dtype, const double *p, int p_mode,
                          int eos, int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS,
                          int mixed_ok) {
  if (!T || !S) return fail(MLX_E_NULL, "T and S must not be NULL");
  if (dtype == MLX_DTYPE_T32_S64 || dtype == MLX_DTYPE_T64_S32) {
    if (!mixed_ok) return fail(MLX_E_ENUM, "theta/salinity of different dtypes: use mlx_eos_map_promote");
  } else if (dtype != MLX_DTYPE_F64 && dtype != MLX_DTYPE_F32 && dtype != MLX_DTYPE_F32_UPCAST)
    return fail(MLX_E_ENUM, "dtype must be one of MLX_DTYPE_*");
  if (eos != MLX_EOS_WRIGHT && eos != MLX_EOS_LINEAR) return fail(MLX_E_ENUM, "unknown eos");
  if (p_mode < MLX_P_SCALAR || p_mode > MLX_P_FULL4D) return fail(MLX_E_ENUM, "unknown p_mode");
  if (!p && eos == MLX_EOS_WRIGHT) return fail(MLX_E_NULL, "p must not be NULL for the Wright EOS");
  if (!p && p_mode != MLX_P_SCALAR) return fail(MLX_E_NULL, "a NULL p requires p_mode MLX_P_SCALAR");
  if (nt <= 0 || nz <= 0 || plane <= 0) return fail(MLX_E_SHAPE, "nt, nz, plane must be > 0");
  if (sT < 0 || sS < 0) return fail(MLX_E_SHAPE, "time strides must be >= 0");
  return 0;
}
/* the pointwise maps: no mixed theta/salinity dtypes (mlx_eos_map_promote's job) */
static int check_common(const void *T, const void *S, int dtype, const double *p, int p_mode,
                        int eos, int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS) {
  return check_common_m(T, S, dtype, p, p_mode, eos, nt, nz, plane, sT, sS, 0);
}

int mlx_version(void) { return MLX_ABI_VERSION; }

int mlx_build_kind(void) { return MLX_BUILD_HOST; } /* the product refuses to bind this build */

int mlx_last_error(char *buf, size_t n) {
  if (buf && n) {
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(g_err);
}

int mlx_last_kernel(char *buf, size_t n) { /* the host build launches no kernels */
  if (buf && n) buf[0] = 0;
  return 0;
}

/* ---- K0 ------------------------------------------------------------------------------------- */
static int eos_map_impl(const void *T, const void *S, int dtype, const double *p, int p_mode,
                        int eos, int func, double aux, int64_t nt, int64_t nz, int64_t plane,
                        int64_t sT, int64_t sS, int flags, double *out) {
  if (flags & MLX_FLAG_FMA) return fail(MLX_E_ENUM, "the host build computes exact arithmetic only");
  if (flags) return fail(MLX_E_ENUM, "mlx_eos_map takes MLX_FLAG_FMA only");
  int rc = check_common(T, S, dtype, p, p_mode, eos, nt, nz, plane, sT, sS);
  if (rc) return rc;
  if (!out) return fail(MLX_E_NULL, "out must not be NULL");
  if (func < MLX_FUNC_DENSITY || func > MLX_FUNC_IBH) return fail(MLX_E_ENUM, "unknown func");
  const double zero = 0.0;
  const double *pp = p ? p : &zero;
  const int pm = p ? p_mode : MLX_P_SCALAR;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t t = 0; t < nt; ++t)
    for (int64_t z = 0; z < nz; ++z)
      for (int64_t i = 0; i < plane; ++i) {
        const Val a = load(T, t * sT + z * plane + i, field_dtype(dtype, 1)), b = load(S, t * sS + z * plane + i, field_dtype(dtype, 0));
        out[(t * nz + z) * plane + i] = eos_eval(eos, func, a, b, pressure(pp, pm, t, z, i, nz, plane), aux);
      }
  return 0;
}

int mlx_eos_map(const void *T, const void *S, int dtype, const double *p, int p_mode, int eos,
                int func, int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS, int flags,
                double *out, void *stream) {
  (void)stream;
  if (func == MLX_FUNC_IBH) return fail(MLX_E_ENUM, "use mlx_inverse_barometer for MLX_FUNC_IBH");
  return eos_map_impl(T, S, dtype, p, p_mode, eos, func, 0.0, nt, nz, plane, sT, sS, flags, out);
}

int mlx_inverse_barometer(const void *T, const void *S, int dtype, const double *p, int p_mode,
                          int eos, double gravity, int64_t nt, int64_t nz, int64_t plane,
                          int64_t sT, int64_t sS, double *out, void *stream) {
  (void)stream;
  if (!p) return fail(MLX_E_NULL, "p must not be NULL");
  return eos_map_impl(T, S, dtype, p, p_mode, eos, MLX_FUNC_IBH, gravity, nt, nz, plane, sT, sS, 0, out);
}

/* ---- K1 ------------------------------------------------------------------------------------- */
size_t mlx_steric_global_workspace_bytes(int64_t nt, int64_t nz, int64_t plane) {
  (void)nt; (void)nz; (void)plane;
  return 0; /* the host build needs no scratch */
}
size_t mlx_steric_global_decomp_workspace_bytes(int64_t nt, int64_t nz, int64_t plane) {
  (void)nt; (void)nz; (void)plane;
  return 0;
}

/* rows: 1 (T, S as given; a stride of 0 holds that field) or 4 (steric, thermo, halo, heat) */
static int global_impl(const void *T, const void *S, const void *T0, const void *S0, int rows,
                       int dtype, const double *vol0, const double *p, int p_mode, int eos,
                       int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS, int flags,
                       double *out) {
  if (flags & MLX_FLAG_FMA) return fail(MLX_E_ENUM, "the host build computes exact arithmetic only");
  if (flags & ~(MLX_FLAG_SKIP_DRY | MLX_FLAG_TCHUNK_MASK)) return fail(MLX_E_ENUM, "unknown flag bits");
  int rc = check_common_m(T, S, dtype, p, p_mode, eos, nt, nz, plane, sT, sS, 1);
  if (rc) return rc;
  if (!vol0 || !out) return fail(MLX_E_NULL, "vol0 and the output must not be NULL");
  if (rows == 4 && (!T0 || !S0)) return fail(MLX_E_NULL, "T0 and S0 must not be NULL");
  const double zero = 0.0;
  const double *pp = p ? p : &zero;
  const int pm = p ? p_mode : MLX_P_SCALAR;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t t = 0; t < nt; ++t) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t z = 0; z < nz; ++z)
      for (int64_t i = 0; i < plane; ++i) {
        const int64_t c = z * plane + i;
        const double v = vol0[c];
        const double pr = pressure(pp, pm, t, z, i, nz, plane);
        const Val a = load(T, t * sT + c, field_dtype(dtype, 1)), b = load(S, t * sS + c, field_dtype(dtype, 0));
        acc[0] += nan0(eos_eval(eos, MLX_FUNC_DENSITY, a, b, pr, 0.0) * v); /* derived.py:435 */
        if (rows == 4) {
          const Val a0 = load(T0, c, field_dtype(dtype, 1)), b0 = load(S0, c, field_dtype(dtype, 0));
          acc[1] += nan0(eos_eval(eos, MLX_FUNC_DENSITY, a, b0, pr, 0.0) * v);
          acc[2] += nan0(eos_eval(eos, MLX_FUNC_DENSITY, a0, b, pr, 0.0) * v);
          acc[3] += nan0(a.d * v);
        }
      }
    for (int r = 0; r < rows; ++r) out[r * nt + t] = acc[r];
  }
  return 0;
}

int mlx_steric_global(const void *T, const void *S, int dtype, const double *vol0, const double *p,
                      int p_mode, int eos, int64_t nt, int64_t nz, int64_t plane, int64_t sT,
                      int64_t sS, int flags, double *masso_out, void *workspace,
                      size_t workspace_bytes, void *stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  return global_impl(T, S, NULL, NULL, 1, dtype, vol0, p, p_mode, eos, nt, nz, plane, sT, sS, flags,
                     masso_out);
}

int mlx_steric_global_decomp(const void *T, const void *S, const void *T0, const void *S0,
                             int dtype, const double *vol0, const double *p, int p_mode, int eos,
                             int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS,
                             int flags, double *out, void *workspace, size_t workspace_bytes,
                             void *stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  if (sT == 0 || sS == 0)
    return fail(MLX_E_SHAPE, "mlx_steric_global_decomp streams both fields: time strides must be > 0");
  return global_impl(T, S, T0, S0, 4, dtype, vol0, p, p_mode, eos, nt, nz, plane, sT, sS, flags, out);
}

/* ---- K2 ------------------------------------------------------------------------------------- */
int mlx_fold_mask(const double *rho0, const double *vol0, int64_t n, double *rho0m_out, void *stream) {
  (void)stream;
  if (!rho0 || !vol0 || !rho0m_out) return fail(MLX_E_NULL, "rho0, vol0, rho0m_out must not be NULL");
  if (n <= 0) return fail(MLX_E_SHAPE, "n must be > 0");
  for (int64_t i = 0; i < n; ++i) rho0m_out[i] = (vol0[i] == vol0[i]) ? rho0[i] : canonical_nan();
  return 0;
}

static inline double dz_default(double depth, double ztop, double zbot) { /* derived.py:295-318 */
  const double d = (depth == depth) ? depth : 0.0;
  const double dz_field = zbot - ztop;
  double part = d - ztop;
  part = (part < 0.0) ? 0.0 : part;
  double result = (part < dz_field) ? part : dz_field;
  part = zbot - 0.0;
  part = (part < 0.0) ? 0.0 : part;
  return (part < result) ? part : result;
}

static int local_impl(const void *T, const void *S, const void *T0, const void *S0, int nvar,
                      int dtype, const double *rho0m, const double *surf, const double *dz,
                      const double *z_i, const double *deptho, const double *p, int p_mode, int eos,
                      double neg_inv_rhozero, int64_t nt, int64_t nz, int64_t plane, int64_t sT,
                      int64_t sS, int flags, double *drho, int64_t drho_vs, double *eta,
                      int64_t eta_vs) {
  if (flags & MLX_FLAG_FMA) return fail(MLX_E_ENUM, "the host build computes exact arithmetic only");
  if (flags & ~MLX_FLAG_SKIP_DRY) return fail(MLX_E_ENUM, "unknown flag bits");
  int rc = check_common_m(T, S, dtype, p, p_mode, eos, nt, nz, plane, sT, sS, 1);
  if (rc) return rc;
  if (!rho0m || !surf || !eta) return fail(MLX_E_NULL, "rho0m, vol0_surface and eta_out must not be NULL");
  if (!dz && (!z_i || !deptho)) return fail(MLX_E_NULL, "either dz or both z_i and deptho must be given");
  if (nvar == 3 && (!T0 || !S0)) return fail(MLX_E_NULL, "T0 and S0 must not be NULL");
  const double zero = 0.0;
  const double *pp = p ? p : &zero;
  const int pm = p ? p_mode : MLX_P_SCALAR;
  const int64_t n3 = nz * plane;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t t = 0; t < nt; ++t)
    for (int64_t i = 0; i < plane; ++i) {
      double acc[3] = {0.0, 0.0, 0.0};
      for (int64_t z = 0; z < nz; ++z) { /* ascending, from +0.0: numpy's axis reduce */
        const int64_t c = z * plane + i;
        const double dzc = dz ? dz[c] : dz_default(deptho[i], z_i[z], z_i[z + 1]);
        const double pr = pressure(pp, pm, t, z, i, nz, plane);
        const Val a = load(T, t * sT + c, field_dtype(dtype, 1)), b = load(S, t * sS + c, field_dtype(dtype, 0));
        double rho[3];
        rho[0] = eos_eval(eos, MLX_FUNC_DENSITY, a, b, pr, 0.0);
        if (nvar == 3) {
          const Val a0 = load(T0, c, field_dtype(dtype, 1)), b0 = load(S0, c, field_dtype(dtype, 0));
          rho[1] = eos_eval(eos, MLX_FUNC_DENSITY, a, b0, pr, 0.0);
          rho[2] = eos_eval(eos, MLX_FUNC_DENSITY, a0, b, pr, 0.0);
        }
        for (int v = 0; v < nvar; ++v) {
          double dr = rho[v] - rho0m[c]; /* steric.py:152 */
          if (dr != dr) dr = canonical_nan();
          if (drho) drho[v * drho_vs + t * n3 + c] = dr;
          acc[v] += nan0(dzc * dr); /* steric.py:163 */
        }
      }
      for (int v = 0; v < nvar; ++v)
        eta[v * eta_vs + t * plane + i] = (surf[i] == surf[i]) ? neg_inv_rhozero * acc[v] : canonical_nan();
    }
  return 0;
}

int mlx_steric_local(const void *T, const void *S, int dtype, const double *rho0m,
                     const double *vol0_surface, const double *dz, const double *z_i,
                     const double *deptho, const double *p, int p_mode, int eos,
                     double neg_inv_rhozero, int64_t nt, int64_t nz, int64_t plane, int64_t sT,
                     int64_t sS, int flags, double *delta_rho_out, double *eta_out, void *stream) {
  (void)stream;
  return local_impl(T, S, NULL, NULL, 1, dtype, rho0m, vol0_surface, dz, z_i, deptho, p, p_mode, eos,
                    neg_inv_rhozero, nt, nz, plane, sT, sS, flags, delta_rho_out, 0, eta_out, 0);
}

int mlx_steric_local_decomp(const void *T, const void *S, const void *T0, const void *S0, int dtype,
                            const double *rho0m, const double *vol0_surface, const double *dz,
                            const double *z_i, const double *deptho, const double *p, int p_mode,
                            int eos, double neg_inv_rhozero, int64_t nt, int64_t nz, int64_t plane,
                            int64_t sT, int64_t sS, int flags, double *delta_rho_out,
                            int64_t delta_rho_variant_stride, double *eta_out,
                            int64_t eta_variant_stride, void *stream) {
  (void)stream;
  if (sT == 0 || sS == 0)
    return fail(MLX_E_SHAPE, "mlx_steric_local_decomp streams both fields: time strides must be > 0");
  if (nt > 0 && nz > 0 && plane > 0 &&
      (eta_variant_stride < nt * plane || (delta_rho_out && delta_rho_variant_stride < nt * nz * plane)))
    return fail(MLX_E_SHAPE, "variant strides must be >= the size of one variant's field");
  return local_impl(T, S, T0, S0, 3, dtype, rho0m, vol0_surface, dz, z_i, deptho, p, p_mode, eos,
                    neg_inv_rhozero, nt, nz, plane, sT, sS, flags, delta_rho_out,
                    delta_rho_variant_stride, eta_out, eta_variant_stride);
}

/* ---- sums, helpers --------------------------------------------------------------------------- */
size_t mlx_nansum_workspace_bytes(int64_t n) { (void)n; return 0; }

int mlx_nansum(const double *x, int64_t n, double *out, void *workspace, size_t workspace_bytes,
               void *stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  if (!x || !out) return fail(MLX_E_NULL, "x and out must not be NULL");
  if (n <= 0) return fail(MLX_E_SHAPE, "n must be > 0");
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) acc += nan0(x[i]);
  out[0] = acc;
  return 0;
}

int mlx_masso(const double *rho, const double *vol, int64_t nt, int64_t n3, int64_t vol_t_stride,
              double *masso_out, void *workspace, size_t workspace_bytes, void *stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  if (!rho || !vol || !masso_out) return fail(MLX_E_NULL, "rho, vol, masso_out must not be NULL");
  if (nt <= 0 || n3 <= 0) return fail(MLX_E_SHAPE, "need nt > 0, n3 > 0");
  if (vol_t_stride != 0 && vol_t_stride != n3) return fail(MLX_E_SHAPE, "vol_t_stride must be 0 or n3");
  for (int64_t t = 0; t < nt; ++t) {
    double acc = 0.0;
    for (int64_t i = 0; i < n3; ++i) acc += nan0(rho[t * n3 + i] * vol[t * vol_t_stride + i]);
    masso_out[t] = acc;
  }
  return 0;
}

int mlx_group_weighted_mean(const double *x, const double *w, int64_t ngroups, int64_t group_len,
                            int64_t n, double *out, void *stream) {
  (void)stream;
  if (!x || !w || !out) return fail(MLX_E_NULL, "x, w, out must not be NULL");
  if (ngroups <= 0 || group_len <= 0 || n <= 0) return fail(MLX_E_SHAPE, "extents must be > 0");
  for (int64_t g = 0; g < ngroups; ++g)
    for (int64_t i = 0; i < n; ++i) {
      double num = 0.0, den = 0.0; /* util.py:85-92: NaNs carry no weight, terms in time order */
      for (int64_t j = 0; j < group_len; ++j) {
        const double v = x[(g * group_len + j) * n + i], wj = w[g * group_len + j];
        num += ((v == v) ? v : 0.0) * wj;
        den += ((v == v) ? 1.0 : 0.0) * wj;
      }
      out[g * n + i] = num / ((den != 0.0) ? den : canonical_nan());
    }
  return 0;
}

int mlx_calc_dz(const double *z_i, const double *depth, int64_t nz, int64_t plane, double top,
                double bottom, int has_bottom, int fraction, double *dz_out, void *stream) {
  (void)stream;
  if (!z_i || !depth || !dz_out) return fail(MLX_E_NULL, "z_i, depth, dz_out must not be NULL");
  if (nz <= 0 || plane <= 0) return fail(MLX_E_SHAPE, "nz and plane must be > 0");
  for (int64_t i = 0; i < plane; ++i) { /* derived.py:295-323 */
    double d = depth[i];
    d = (d == d) ? d : 0.0;
    if (has_bottom) d = (bottom != bottom || bottom < d) ? bottom : d; /* np.minimum */
    for (int64_t z = 0; z < nz; ++z) {
      const double ztop = z_i[z], zbot = z_i[z + 1], dz_field = zbot - ztop;
      double part = d - ztop;
      part = (part < 0.0) ? 0.0 : part;
      double result = (part != part || part < dz_field) ? part : dz_field;
      part = zbot - top;
      part = (part < 0.0) ? 0.0 : part;
      result = (part != part || part < result) ? part : result;
      if (fraction) {
        const double f = (dz_field == 0.0) ? canonical_nan() : dz_field;
        const double g = (result == 0.0) ? canonical_nan() : result;
        result = g / f;
      }
      dz_out[z * plane + i] = result;
    }
  }
  return 0;
}

int mlx_stream_probe(const double *a, const double *b, int64_t n, double *out, void *stream) {
  (void)stream;
  if (!a || !b || !out) return fail(MLX_E_NULL, "a, b, out must not be NULL");
  if (n <= 0 || n % 2) return fail(MLX_E_SHAPE, "n must be > 0 and even");
  for (int64_t i = 0; i < n; ++i) out[i] = a[i] + b[i];
  return 0;
}

int mlx_stream_probe_mix(const void *a, const void *b, int dtype, int64_t n, double *out,
                         int write_out, void *stream) {
  (void)stream;
  if (!a || !out) return fail(MLX_E_NULL, "a and out must not be NULL");
  if (dtype != MLX_DTYPE_F64 && dtype != MLX_DTYPE_F32)
    return fail(MLX_E_ENUM, "dtype must be MLX_DTYPE_F64 or MLX_DTYPE_F32");
  if (n <= 0 || n % (dtype == MLX_DTYPE_F64 ? 2 : 4))
    return fail(MLX_E_SHAPE, "n must be > 0 and a whole number of 16-byte packs");
  if (!write_out) return 0; /* the read-only probe leaves nothing behind */
  for (int64_t i = 0; i < n; ++i) {
    double r = (dtype == MLX_DTYPE_F64) ? ((const double *)a)[i] : (double)((const float *)a)[i];
    if (b) r += (dtype == MLX_DTYPE_F64) ? ((const double *)b)[i] : (double)((const float *)b)[i];
    out[i] = r;
  }
  return 0;
}

int mlx_valu_probe(int64_t iters, double *out, int64_t *lane_instructions, void *stream) {
  (void)stream; /* a device measurement aid: nothing to run on the host */
  if (!out || !lane_instructions) return fail(MLX_E_NULL, "out and lane_instructions must not be NULL");
  if (iters <= 0 || iters > (1 << 24)) return fail(MLX_E_SHAPE, "need 0 < iters <= 2^24");
  *lane_instructions = (int64_t)8192 * 256 * 8 * iters;
  return 0;
}

/* ---- stratification diagnostics (derived.py:328-411, :714-766, :30-71, :798-831) ---------- */
static inline double gradient_at(const void *f, int dtype, int64_t i0, int64_t stride, double a,
                                 double b, double c, int central, double two_dx) {
  /* numpy.gradient, one value: levels i0, i0+stride, i0+2*stride */
  if (dtype == MLX_DTYPE_F32) { /* float32 array: float64 products, float32 result */
    const float *x = (const float *)f;
    const float f0 = x[i0], f1 = x[i0 + stride], f2 = x[i0 + 2 * stride];
    if (central) return (double)(float)((double)(f2 - f0) / two_dx);
    return (double)(float)((a * (double)f0 + b * (double)f1) + c * (double)f2);
  }
  double d0, d1, d2;
  if (dtype == MLX_DTYPE_F64) {
    const double *x = (const double *)f;
    d0 = x[i0], d1 = x[i0 + stride], d2 = x[i0 + 2 * stride];
  } else {
    const float *x = (const float *)f;
    d0 = (double)x[i0], d1 = (double)x[i0 + stride], d2 = (double)x[i0 + 2 * stride];
  }
  if (central) return (d2 - d0) / two_dx;
  return (a * d0 + b * d1) + c * d2;
}

int mlx_stratification(const void *T, const void *S, int dtype, const double *p,
                       int64_t p_stride_t, int64_t p_stride_z, int64_t p_stride_cell, int eos,
                       int func, const double *coef, int uniform, double two_dx, double gravity,
                       int64_t nt, int64_t nz, int64_t plane, double *out, void *stream) {
  (void)stream;
  if (!T || !S || !coef || !out) return fail(MLX_E_NULL, "T, S, coef and out must not be NULL");
  if (eos != MLX_EOS_WRIGHT && eos != MLX_EOS_LINEAR) return fail(MLX_E_ENUM, "unknown eos");
  if (func != MLX_STRAT_N2 && func != MLX_STRAT_TURNER) return fail(MLX_E_ENUM, "unknown func");
  if (dtype != MLX_DTYPE_F64 && dtype != MLX_DTYPE_F32 && dtype != MLX_DTYPE_F32_UPCAST)
    return fail(MLX_E_ENUM, "dtype must be MLX_DTYPE_F64, _F32 or _F32_UPCAST");
  if (eos == MLX_EOS_LINEAR && dtype == MLX_DTYPE_F32)
    return fail(MLX_E_ENUM, "linear EOS on float32 fields is float32 throughout in numpy: not built");
  if (!p && eos == MLX_EOS_WRIGHT) return fail(MLX_E_NULL, "p must not be NULL for the Wright EOS");
  if (nt <= 0 || plane <= 0) return fail(MLX_E_SHAPE, "nt and plane must be > 0");
  if (nz < 3) return fail(MLX_E_SHAPE, "nz must be >= 3 (second-order edges)");
  if (p_stride_t < 0 || p_stride_z < 0 || p_stride_cell < 0 || p_stride_cell > 1)
    return fail(MLX_E_SHAPE, "pressure strides must be >= 0 (cell stride 0 or 1)");
  if (uniform && !(two_dx == two_dx && two_dx != 0.0))
    return fail(MLX_E_SHAPE, "uniform spacing needs a non-zero two_dx");
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t t = 0; t < nt; ++t)
    for (int64_t k = 0; k < nz; ++k) {
      const int64_t k0 = (k == 0) ? 0 : (k == nz - 1) ? nz - 3 : k - 1; /* first level of the stencil */
      const int central = uniform && k > 0 && k < nz - 1;
      const double a = coef[3 * k], b = coef[3 * k + 1], c = coef[3 * k + 2];
      for (int64_t i = 0; i < plane; ++i) {
        const int64_t at = (t * nz + k) * plane + i, at0 = (t * nz + k0) * plane + i;
        const double pk = p ? p[t * p_stride_t + k * p_stride_z + i * p_stride_cell] : 0.0;
        const Val Tv = load(T, at, dtype), Sv = load(S, at, dtype);
        const double alpha = eos_eval(eos, MLX_FUNC_ALPHA, Tv, Sv, pk, 0.0);
        const double beta = eos_eval(eos, MLX_FUNC_BETA, Tv, Sv, pk, 0.0);
        const double dtdz = gradient_at(T, dtype, at0, plane, a, b, c, central, two_dx);
        const double dsdz = gradient_at(S, dtype, at0, plane, a, b, c, central, two_dx);
        if (func == MLX_STRAT_N2) {
          out[at] = gravity * ((alpha * dtdz) - (beta * dsdz));
        } else {
          const double r = (beta * dsdz) / (alpha * dtdz);
          out[at] = atan((1.0 + r) / (1.0 - r)) * (180.0 / 3.14159265358979323846);
        }
      }
    }
  return 0;
}

int mlx_adjust_negative_n2(const double *n2, int64_t nt, int64_t nz, int64_t plane,
                           int64_t lead0_rows, const double *dz, double *adjusted, double *speed,
                           void *stream) {
  (void)stream;
  if (!n2) return fail(MLX_E_NULL, "n2 must not be NULL");
  if (!adjusted && !speed) return fail(MLX_E_NULL, "one of adjusted / speed is required");
  if (speed && !dz) return fail(MLX_E_NULL, "speed needs dz");
  if (nt <= 0 || nz <= 0 || plane <= 0) return fail(MLX_E_SHAPE, "nt, nz, plane must be > 0");
  if (lead0_rows < 0 || lead0_rows > nt || (lead0_rows == 0 && nt != 1))
    return fail(MLX_E_SHAPE, "lead0_rows must be in 1..nt, or 0 with nt == 1");
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t t = 0; t < nt; ++t)
    for (int64_t i = 0; i < plane; ++i) {
      double carried = NAN, sum = 0.0;
      for (int64_t k = 0; k < nz; ++k) {
        const double x = n2[(t * nz + k) * plane + i];
        double a = (x <= 0.0) ? NAN : x;
        const int lead0 = lead0_rows ? (t < lead0_rows) : (k == 0);
        if (lead0 && a != a) a = 1.0e-8;
        if (a != a) a = carried;
        carried = a;
        const double masked = (x != x) ? NAN : a;
        if (adjusted) adjusted[(t * nz + k) * plane + i] = masked;
        if (speed) {
          const double term = sqrt(masked) * dz[k * plane + i];
          if (term == term) sum += term;
        }
      }
      if (speed) {
        const double surface = n2[t * nz * plane + i];
        speed[t * plane + i] = (!lead0_rows && surface != surface) ? NAN : sum / 3.14159265358979323846;
      }
    }
  return 0;
}

int mlx_wave_speed_where_time0(const double *n2_t0, const double *speed, int64_t nt, int64_t nz,
                               int64_t plane, double *out, void *stream) {
  (void)stream;
  if (!n2_t0 || !speed || !out) return fail(MLX_E_NULL, "n2_t0, speed and out must not be NULL");
  if (nt <= 0 || nz <= 0 || plane <= 0) return fail(MLX_E_SHAPE, "nt, nz, plane must be > 0");
  for (int64_t kc = 0; kc < nz * plane; ++kc)
    for (int64_t t = 0; t < nt; ++t)
      out[kc * nt + t] = (n2_t0[kc] != n2_t0[kc]) ? NAN : speed[t * plane + kc % plane];
  return 0;
}

int mlx_host_copy(void *dst, const void *src, size_t nbytes, int threads, int streaming) {
  (void)streaming; /* the checker's build has no tuned copy: one plain memcpy, same contract */
  if (nbytes == 0) return 0;
  if (!dst || !src) return fail(MLX_E_NULL, "dst and src must not be NULL");
  if (threads < 1 || threads > 64) return fail(MLX_E_SHAPE, "threads must be in 1..64");
  const unsigned char *s = (const unsigned char *)src;
  unsigned char *d = (unsigned char *)dst;
  if (d < s + nbytes && s < d + nbytes) return fail(MLX_E_SHAPE, "dst and src overlap");
  memcpy(d, s, nbytes);
  return 0;
}

int mlx_host_copy_masked(void *dst, const void *src, const unsigned char *mask, size_t n,
                         int elem_size, int threads) {
  /* the checker's build: one plain loop, same contract (include/momlevel_hip.h) */
  if (n == 0) return 0;
  if (!dst || !src || !mask) return fail(MLX_E_NULL, "dst, src and mask must not be NULL");
  if (elem_size != 4 && elem_size != 8) return fail(MLX_E_ENUM, "elem_size must be 4 or 8");
  if (threads < 1 || threads > 64) return fail(MLX_E_SHAPE, "threads must be in 1..64");
  if (((uintptr_t)dst | (uintptr_t)src) % (size_t)elem_size)
    return fail(MLX_E_ALIGN, "dst / src not element-aligned");
  const unsigned char *s = (const unsigned char *)src, *d = (const unsigned char *)dst;
  const size_t nbytes = n * (size_t)elem_size;
  if ((d < s + nbytes && s < d + nbytes) || (d < mask + n && mask < d + nbytes))
    return fail(MLX_E_SHAPE, "dst overlaps src or mask");
  if (elem_size == 4) {
    uint32_t *o = (uint32_t *)dst;
    const uint32_t *i4 = (const uint32_t *)src;
    for (size_t i = 0; i < n; ++i) o[i] = mask[i] ? 0x7FC00000u : i4[i];
  } else {
    uint64_t *o = (uint64_t *)dst;
    const uint64_t *i8 = (const uint64_t *)src;
    for (size_t i = 0; i < n; ++i) o[i] = mask[i] ? 0x7FF8000000000000ull : i8[i];
  }
  return 0;
}

static inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

int mlx_synth_field(void *out, int dtype, int64_t nt, int64_t nz, int64_t ny, int64_t nx,
                    int64_t t0, int64_t NY, int64_t NX, int64_t y0, int64_t x0, uint64_t seed,
                    int field_id, double lo, double scale, const double *mask3d, void *stream) {
  (void)stream;
  if (!out) return fail(MLX_E_NULL, "out must not be NULL");
  if (nt <= 0 || nz <= 0 || ny <= 0 || nx <= 0) return fail(MLX_E_SHAPE, "dims must be > 0");
  if (y0 < 0 || x0 < 0 || t0 < 0 || ny > NY || nx > NX || y0 > NY - ny || x0 > NX - nx)
    return fail(MLX_E_SHAPE, "tile does not fit the global grid");
  if (field_id < 0 || field_id > 15) return fail(MLX_E_ENUM, "field_id must be 0..15");
  if (dtype != MLX_DTYPE_F64 && dtype != MLX_DTYPE_F32 && dtype != MLX_DTYPE_F32_UPCAST)
    return fail(MLX_E_ENUM, "dtype must be MLX_DTYPE_F64, _F32 or _F32_UPCAST");
  for (int64_t t = 0; t < nt; ++t)
    for (int64_t z = 0; z < nz; ++z)
      for (int64_t y = 0; y < ny; ++y)
        for (int64_t x = 0; x < nx; ++x) {
          const uint64_t g = (uint64_t)((((t0 + t) * nz + z) * NY + (y0 + y)) * NX + (x0 + x));
          const uint64_t h = splitmix64(seed ^ ((uint64_t)field_id << 60) ^ g);
          double v = lo + scale * ((double)(h >> 11) * 0x1.0p-53);
          const int64_t r = (z * ny + y) * nx + x;
          if (mask3d && mask3d[r] != mask3d[r]) v = canonical_nan();
          if (dtype == MLX_DTYPE_F64) ((double *)out)[t * nz * ny * nx + r] = v;
          else ((float *)out)[t * nz * ny * nx + r] = (float)v;
        }
  return 0;
}
