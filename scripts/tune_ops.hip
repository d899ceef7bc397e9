// tune_ops.hip -- issue cost of individual fp64 VALU instructions on gfx950 (wave-cycles per
// instruction per SIMD at full occupancy).  Developer microbenchmark, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP %s\n",hipGetErrorString(e_));exit(1);} } while(0)

template <int OP>
__global__ __launch_bounds__(256) void kop(const double* in, double* out, int iters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = in[i * 8 + k];
  const double b = in[0] * 1.0000001, c = in[1];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
      if constexpr (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
      if constexpr (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      if constexpr (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
      if constexpr (OP == 4) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a[k]) : "v"(b) : "vcc");
      if constexpr (OP == 5) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
      if constexpr (OP == 6) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      if constexpr (OP == 7) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a[k])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f)); }
      if constexpr (OP == 8) { float f = (float)a[k]; asm volatile("v_rcp_f32 %0, %0" : "+v"(f)); a[k] = (double)f; }
      if constexpr (OP == 9) asm volatile("v_cmp_o_f64 vcc, %0, %0\n\tv_cndmask_b32 %1, 0, %1, vcc" : : "v"(a[k]), "v"(*(int*)&a[k]) : "vcc");
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  out[i] = s;
}

int main() {
  const int nblk = 256 * 8;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
  double *in, *out;
  CK(hipMalloc(&in, (size_t)nblk * 256 * 8 * 8)); CK(hipMalloc(&out, (size_t)nblk * 256 * 8));
  double* h = (double*)malloc((size_t)nblk * 256 * 8 * 8);
  for (size_t i = 0; i < (size_t)nblk * 256 * 8; ++i) h[i] = 1.0 + (i % 97) * 1e-3;
  CK(hipMemcpy(in, h, (size_t)nblk * 256 * 8 * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  const char* names[10] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_rcp_f64", "v_div_scale_f64",
                           "v_div_fmas_f64", "v_div_fixup_f64", "cvt f64->f32->f64 (2 instr)",
                           "cvt+v_rcp_f32+cvt (3 instr)", "v_cmp_o_f64+v_cndmask (2 instr)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int op = 0; op < 10; ++op) {
      CK(hipEventRecord(e0));
      switch (op) {
        case 0: hipLaunchKernelGGL(kop<0>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 1: hipLaunchKernelGGL(kop<1>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 2: hipLaunchKernelGGL(kop<2>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 3: hipLaunchKernelGGL(kop<3>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 4: hipLaunchKernelGGL(kop<4>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 5: hipLaunchKernelGGL(kop<5>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 6: hipLaunchKernelGGL(kop<6>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 7: hipLaunchKernelGGL(kop<7>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 8: hipLaunchKernelGGL(kop<8>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
        case 9: hipLaunchKernelGGL(kop<9>, dim3(nblk), dim3(256), 0, 0, in, out, iters); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      // wave-instructions per SIMD: 8 waves/SIMD * iters * 8 ops
      const double per_simd = 8.0 * iters * 8;
      if (rep) printf("%-34s %8.3f ms   %6.2f ns per wave-op per SIMD  (~%5.1f cycles at 2.1 GHz)\n",
                      names[op], ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.1);
    }
  return 0;
}
