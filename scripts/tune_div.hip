// tune_div.hip -- cost of a GUARDED scale-free reciprocal inside the exact Wright density
// (round 2; follows scripts/tune_alu.hip).  The scale-free sequence is hipcc's own IEEE f64
// division expansion minus v_div_scale / v_div_fmas / v_div_fixup, so it is bit-identical whenever
// no scaling would have happened; the guard sends everything else (exponent outside 2^+-900, inf,
// NaN, 0, denormals) to the IEEE division.  Variants: how the guard is evaluated.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/tune_div.hip -o scripts/tune_div
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../momlevel_amd/csrc/eos_device.hpp"
#pragma clang fp contract(off)
using namespace mlx;
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP %s\n",hipGetErrorString(e_));exit(1);} } while(0)

__device__ __forceinline__ double rcp_noscale(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(e, r, r);
}
// exponent field strictly inside (1023-900, 1023+900): one and, one sub, one unsigned compare on the
// HIGH dword only
__device__ __forceinline__ bool in_range(double x) {
  const unsigned hi = (unsigned)__double2hiint(x) & 0x7FFFFFFFu;
  return (hi - ((1023u - 900u) << 20)) < (1800u << 20);
}
template <int MATH>
__device__ __forceinline__ void rho4(const double* T, const double* S, double p, double* rho) {
  double pp0[4], den[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double al0, p0, lam;
    wright_terms<double>(T[k], S[k], al0, p0, lam);
    pp0[k] = p + p0;
    den[k] = lam + al0 * pp0[k];
  }
  double I[4];
  if constexpr (MATH == 0) {  // IEEE
#pragma unroll
    for (int k = 0; k < 4; ++k) I[k] = 1.0 / den[k];
  } else if constexpr (MATH == 1) {  // unguarded scale-free
#pragma unroll
    for (int k = 0; k < 4; ++k) I[k] = rcp_noscale(den[k]);
  } else if constexpr (MATH == 2) {  // per-cell guard, plain branch
#pragma unroll
    for (int k = 0; k < 4; ++k) I[k] = in_range(den[k]) ? rcp_noscale(den[k]) : 1.0 / den[k];
  } else if constexpr (MATH == 3) {  // per-4-cells guard, wave-uniform branch
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      I[k] = rcp_noscale(den[k]);
      ok = ok && in_range(den[k]);
    }
    if (__builtin_amdgcn_ballot_w64(!ok) != 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) I[k] = 1.0 / den[k];
    }
  } else {  // per-4-cells guard, divergent branch
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      I[k] = rcp_noscale(den[k]);
      ok = ok && in_range(den[k]);
    }
    if (__builtin_expect(!ok, 0)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) I[k] = 1.0 / den[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) rho[k] = pp0[k] * I[k];
}
template <int MATH>
__global__ __launch_bounds__(256) void kalu(const double* in, double* out, int iters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double T[4], S[4];
  for (int k = 0; k < 4; ++k) { T[k] = in[i * 8 + k]; S[k] = in[i * 8 + 4 + k]; }
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    double rho[4];
    rho4<MATH>(T, S, 2.0e5 + it, rho);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += rho[k];
  }
  out[i] = acc;
}
// bit-identity of the guarded reciprocal against IEEE 1.0/x on all kinds of bit patterns
__global__ void kcheck(unsigned long long seed, unsigned long long* mism, int per_thread, int mode) {
  unsigned long long x = seed + (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL;
  unsigned long long bad = 0;
  for (int i = 0; i < per_thread; ++i) {
    x = splitmix64(x);
    double v;
    if (mode == 0) {         // any bit pattern
      v = __longlong_as_double((long long)x);
    } else if (mode == 1) {  // the Wright denominator's neighbourhood, 2^10 .. 2^30, random mantissa
      const unsigned long long mant = x & 0xFFFFFFFFFFFFFULL;
      const long long ex = 1023 + 10 + (long long)((x >> 52) % 21);
      v = __longlong_as_double((long long)(((unsigned long long)ex << 52) | mant));
    } else if (mode == 2) {  // hard mantissas (within 255 ulp of a power of two, both sides), 2^-890..2^890
      const long long ex = 1023 - 890 + (long long)((x >> 40) % 1781);
      const unsigned long long k = x & 0xFF;
      const unsigned long long mant = ((x >> 8) & 1) ? (0xFFFFFFFFFFFFFULL - k) : k;
      v = __longlong_as_double((long long)(((x >> 63) << 63) | ((unsigned long long)ex << 52) | mant));
    } else {                 // the guard's edges: exponent within 3 of 1023 +- 900
      const long long ex = ((x >> 60) & 1 ? 1023 + 900 : 1023 - 900) - 3 + (long long)((x >> 52) % 7);
      v = __longlong_as_double((long long)(((x >> 63) << 63) | ((unsigned long long)ex << 52) | (x & 0xFFFFFFFFFFFFFULL)));
    }
    const double a = 1.0 / v;
    const double b = in_range(v) ? rcp_noscale(v) : 1.0 / v;
    if (__double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b)) ++bad;
  }
  if (bad) atomicAdd(mism, bad);
}
int main() {
  const int nblk = 256 * 8 * 4;
  double *in, *out; unsigned long long* mism;
  CK(hipMalloc(&in, nblk * 256 * 8 * 8)); CK(hipMalloc(&out, nblk * 256 * 8)); CK(hipMalloc(&mism, 8));
  double* h = (double*)malloc(nblk * 256 * 8 * 8);
  for (int i = 0; i < nblk * 256; ++i) for (int k = 0; k < 4; ++k) { h[i*8+k] = -2 + 34.0 * ((i * 7 + k) % 1000) / 1000.0; h[i*8+4+k] = 30 + 10.0 * ((i * 13 + k) % 1000) / 1000.0; }
  CK(hipMemcpy(in, h, nblk * 256 * 8 * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 400;
  const char* names[5] = {"IEEE 1.0/x", "scale-free (unguarded)", "guard per cell", "guard per 4, ballot", "guard per 4, divergent"};
  for (int rep = 0; rep < 2; ++rep)
  for (int m = 0; m < 5; ++m) {
    CK(hipEventRecord(e0));
    if (m == 0) hipLaunchKernelGGL(kalu<0>, dim3(nblk), dim3(256), 0, 0, in, out, iters);
    if (m == 1) hipLaunchKernelGGL(kalu<1>, dim3(nblk), dim3(256), 0, 0, in, out, iters);
    if (m == 2) hipLaunchKernelGGL(kalu<2>, dim3(nblk), dim3(256), 0, 0, in, out, iters);
    if (m == 3) hipLaunchKernelGGL(kalu<3>, dim3(nblk), dim3(256), 0, 0, in, out, iters);
    if (m == 4) hipLaunchKernelGGL(kalu<4>, dim3(nblk), dim3(256), 0, 0, in, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double cells = (double)nblk * 256 * 4 * iters;
    if (rep) printf("%-26s %8.3f ms  %8.1f Gcells/s\n", names[m], ms, cells / ms / 1e6);
  }
  const char* modes[4] = {"all bit patterns", "2^10..2^30 random", "hard mantissas 2^-890..2^890", "guard edges"};
  for (int mode = 0; mode < 4; ++mode) {
    CK(hipMemset(mism, 0, 8));
    hipLaunchKernelGGL(kcheck, dim3(4096), dim3(256), 0, 0, 777ULL + mode, mism, 4096, mode);
    unsigned long long bad; CK(hipMemcpy(&bad, mism, 8, hipMemcpyDeviceToHost));
    printf("guarded reciprocal vs IEEE, %-30s: %llu mismatches of %.3e\n", modes[mode], bad, 4096.0 * 256 * 4096);
  }
  return 0;
}
