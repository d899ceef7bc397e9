// check_div.hip -- the guarded scale-free reciprocal of eos_device.hpp (rcp_scale_free + the
// v_cmp_class guard on the seed) against hipcc's IEEE 1.0/x, bit for bit, on every class of bit
// pattern: what ExactFastOps relies on to be bit-identical to ExactOps (= to numpy).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/check_div.hip -o scripts/check_div
//   ./scripts/check_div            (a few seconds on one MI355X; exit code 1 on any mismatch)
//
// A lane is compared wherever the guard did NOT object (its bit of the lane mask is clear) -- in the
// kernels an objection sends the whole wave to the IEEE division, so those lanes are identical by
// construction.  Reported per class: samples, lanes the guard accepted, mismatches among them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../momlevel_amd/csrc/eos_device.hpp"
#pragma clang fp contract(off)
using namespace mlx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s\n", hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ double from_bits(unsigned long long b) { return __longlong_as_double((long long)b); }

__device__ double pattern(int mode, unsigned long long x) {
  const unsigned long long sign = (x >> 63) << 63;
  const unsigned long long mant = x & 0xFFFFFFFFFFFFFULL;
  switch (mode) {
    case 0:  // any bit pattern
      return from_bits(x);
    case 1: {  // the Wright denominator's neighbourhood, 2^10 .. 2^30, random mantissa
      const unsigned long long ex = 1023 + 10 + ((x >> 52) % 21);
      return from_bits((ex << 52) | mant);
    }
    case 2: {  // hard mantissas (within 255 ulp of a power of two, both sides), EVERY normal exponent
      const unsigned long long ex = 1 + ((x >> 40) % 2046);
      const unsigned long long k = x & 0xFF;
      const unsigned long long m = ((x >> 8) & 1) ? (0xFFFFFFFFFFFFFULL - k) : k;
      return from_bits(sign | (ex << 52) | m);
    }
    case 3:  // denormals, random mantissa
      return from_bits(sign | mant);
    case 4: {  // the bands where 1/x crosses the top of the range: x around 2^-1024, 2^-1023, 2^-1022
      const unsigned long long centre[3] = {0x0004000000000000ULL, 0x0008000000000000ULL,
                                            0x0010000000000000ULL};
      const long long d = (long long)((x >> 8) & 0xFFFFFF) - 0x800000;
      return from_bits(sign | (unsigned long long)((long long)centre[x % 3] + d));
    }
    case 5: {  // huge: 1/x denormal or nearly -- exponents 2^1019 .. 2^1023, random mantissa
      const unsigned long long ex = 2042 + ((x >> 52) % 5);
      return from_bits(sign | (ex << 52) | mant);
    }
    case 6: {  // the bands around 2^1021, 2^1022, 2^1023 (1/x at the bottom of the normal range)
      const unsigned long long centre[3] = {0x7FC0000000000000ULL, 0x7FD0000000000000ULL,
                                            0x7FE0000000000000ULL};
      const long long d = (long long)((x >> 8) & 0xFFFFFF) - 0x800000;
      return from_bits(sign | (unsigned long long)((long long)centre[x % 3] + d));
    }
    case 7: {  // tiny normals 2^-1022 .. 2^-1000 (1/x near the top), random mantissa
      const unsigned long long ex = 1 + ((x >> 52) % 23);
      return from_bits(sign | (ex << 52) | mant);
    }
    default: {  // every exponent incl. 0 and 2047, mantissas 0, 1, all-ones, random
      const unsigned long long ex = (x >> 40) % 2048;
      const unsigned long long pick = (x >> 60) & 3;
      const unsigned long long m = pick == 0 ? 0 : pick == 1 ? 1 : pick == 2 ? 0xFFFFFFFFFFFFFULL : mant;
      return from_bits(sign | (ex << 52) | m);
    }
  }
}

// (a) ExactFastOps: the window guard on den, any float64 bit pattern
__global__ void kcheck(unsigned long long seed, unsigned long long* counts, int per_thread, int mode) {
  unsigned long long x = seed + (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL;
  unsigned long long bad = 0, accepted = 0;
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < per_thread; ++i) {
    x = splitmix64(x);
    const double v = pattern(mode, x);
    const double num = 1.0 + (double)(x & 0xFFFF);  // the kernels multiply a numerator in afterwards
    lanemask_t dummy = 0, unsafe = 0;
    const double a = ExactOps::quotient(num, v, dummy);      // num * (1.0 / v), IEEE division
    const double b = ExactFastOps::quotient(num, v, unsafe);  // num * rcp_scale_free(v)
    if ((unsafe >> lane) & 1) continue;  // the guard objected: the kernels redo the wave with IEEE
    ++accepted;
    if (__double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b)) ++bad;
  }
  atomicAdd(&counts[0], bad);
  atomicAdd(&counts[1], accepted);
}

// (b) ExactFastF32Ops: the class guard on the seed, for denominators built the way numpy's float32
// polynomial builds them -- al0, p0, lam ANY float32 bit pattern (as float32 values), p a float64
// pressure that passes p_unsafe() -- den = lam + al0*(p + p0), eos/wright.py:46-47
__device__ double pressure_pattern(int pmode, unsigned long long x) {
  const unsigned long long sign = (x >> 63) << 63;
  const unsigned long long mant = x & 0xFFFFFFFFFFFFFULL;
  switch (pmode) {
    case 0: return 101325.0 + (double)(x % 70000000ULL);  // the ocean: 1e5 .. 7e7 Pa
    case 1: return 0.0;
    case 2: {  // anything the scalar test lets through: 2^-200 .. 2^800, both signs
      const unsigned long long ex = 1023 - 200 + ((x >> 40) % 1001);
      return from_bits(sign | (ex << 52) | mant);
    }
    default: {  // the edges of that range
      const unsigned long long ex = ((x >> 60) & 1) ? 1023 + 800 : 1023 - 200;
      return from_bits(sign | (ex << 52) | (((x >> 59) & 1) ? 0 : mant));
    }
  }
}

__global__ void kcheck_f32(unsigned long long seed, unsigned long long* counts, int per_thread,
                           int pmode) {
  unsigned long long x = seed + (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL;
  unsigned long long bad = 0, accepted = 0, vetoed = 0;
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < per_thread; ++i) {
    x = splitmix64(x);
    const unsigned long long y = splitmix64(x ^ 0xABCDEF);
    // three float32 bit patterns: half of the samples uniform over all patterns, half with small
    // exponents around the Wright values (so that lam + al0*pp0 cancels often)
    float f[3];
    for (int k = 0; k < 3; ++k) {
      unsigned int bits = (unsigned int)((k == 0 ? x : k == 1 ? x >> 32 : y) & 0xFFFFFFFFu);
      if ((y >> 63) & 1) bits = (bits & 0x807FFFFFu) | ((100u + (bits >> 23) % 60u) << 23);
      f[k] = __uint_as_float(bits);
    }
    double p = pressure_pattern(pmode, y);
    if ((y >> 62) & 1) p = -(double)f[1];  // force p + p0 == 0 in a quarter of the samples
    if (ExactFastF32Ops::p_unsafe(p) != 0) { ++vetoed; continue; }  // the kernels take IEEE for this level
    const double pp0 = p + (double)f[1];
    const double den = (double)f[2] + (double)f[0] * pp0;
    lanemask_t dummy = 0, unsafe = 0;
    const double a = ExactOps::quotient(pp0, den, dummy);
    const double b = ExactFastF32Ops::quotient(pp0, den, unsafe);
    if ((unsafe >> lane) & 1) continue;
    ++accepted;
    if (__double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b)) ++bad;
  }
  atomicAdd(&counts[0], bad);
  atomicAdd(&counts[1], accepted);
  atomicAdd(&counts[2], vetoed);
}

int main() {
  unsigned long long* counts;
  CK(hipMalloc(&counts, 24));
  const char* modes[9] = {"all bit patterns", "2^10..2^30 (Wright's denominators)",
                          "hard mantissas, every normal exponent", "denormals",
                          "bands at 2^-1024, 2^-1023, 2^-1022", "2^1019..2^1023",
                          "bands at 2^1021, 2^1022, 2^1023", "2^-1022..2^-1000",
                          "every exponent x special mantissas"};
  unsigned long long total_bad = 0;
  const int per_thread = 4096;
  printf("ExactFastOps (window guard on den) vs IEEE 1.0/den, float64 bit patterns\n");
  for (int mode = 0; mode < 9; ++mode) {
    CK(hipMemset(counts, 0, 24));
    hipLaunchKernelGGL(kcheck, dim3(4096), dim3(256), 0, 0, 991ULL + mode, counts, per_thread, mode);
    CK(hipGetLastError());
    unsigned long long h[3];
    CK(hipMemcpy(h, counts, 24, hipMemcpyDeviceToHost));
    printf("  %-40s: %.3e samples, %.3e accepted by the guard, %llu mismatches\n", modes[mode],
           4096.0 * 256 * per_thread, (double)h[1], h[0]);
    total_bad += h[0];
  }
  const char* pmodes[4] = {"p = 1e5..7e7 Pa", "p = 0", "|p| in 2^-200..2^800", "|p| at the range's edges"};
  printf("ExactFastF32Ops (class guard on the seed) vs IEEE, den = lam + al0*(p + p0) from float32 values\n");
  for (int pm = 0; pm < 4; ++pm) {
    CK(hipMemset(counts, 0, 24));
    hipLaunchKernelGGL(kcheck_f32, dim3(4096), dim3(256), 0, 0, 4242ULL + pm, counts, per_thread, pm);
    CK(hipGetLastError());
    unsigned long long h[3];
    CK(hipMemcpy(h, counts, 24, hipMemcpyDeviceToHost));
    printf("  %-40s: %.3e samples, %.3e vetoed by the pressure test, %.3e accepted by the guard, "
           "%llu mismatches\n", pmodes[pm], 4096.0 * 256 * per_thread, (double)h[2], (double)h[1], h[0]);
    total_bad += h[0];
  }
  printf(total_bad ? "FAILED\n" : "guarded scale-free reciprocal == IEEE division wherever the guards accept: OK\n");
  return total_bad ? 1 : 0;
}
