// tune_probe.hip -- what IS this box's streaming ceiling for each read:write mix of the local pass?
// Round 4's probe (k_stream_probe_mix: grid-stride, ONE 16-byte pack in flight per thread) read
// 4.1 TB/s for "1 x float32 in, 1 x float64 out" while the kernel it was meant to bound ran at
// 5.1 TB/s: a probe slower than the kernel is not a ceiling.  This sweeps the probe's shape:
//   TIn in {double, float} x NIN in {1, 2} x WRITE in {0, 1}
//   U   packs of 16 B in flight per thread and stream (1, 2, 4, 8)
//   grid: blocks per CU (grid-stride over the rest) or one tile per block (no loop)
//   nontemporal stores on/off
//   hipcc --offload-arch=gfx950 -O3 scripts/tune_probe.hip -o scripts/tune_probe
//   ./scripts/tune_probe [GiB of the float64 output stream, default 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP %s at %d\n",hipGetErrorString(e_),__LINE__);exit(1);} } while(0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T, int VEC> struct Pk { T v[VEC]; };
template <typename T, int VEC>
__device__ __forceinline__ Pk<T, VEC> ld(const T* p) {
  f4 r = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  Pk<T, VEC> d; __builtin_memcpy(&d, &r, 16); return d;
}
template <bool NT>
__device__ __forceinline__ void st2(double* p, double a, double b) {
  f4 r; double t[2] = {a, b}; __builtin_memcpy(&r, t, 16);
  if (NT) __builtin_nontemporal_store(r, reinterpret_cast<f4*>(p));
  else *reinterpret_cast<f4*>(p) = r;
}

// LOOP: grid-stride over tiles of 256*U packs; !LOOP: one tile per block
template <typename TIn, int NIN, bool WRITE, int U, bool LOOP, bool NTS>
__global__ __launch_bounds__(256) void k_probe(const TIn* __restrict__ a, const TIn* __restrict__ b,
                                               int64_t npacks, double* __restrict__ out) {
  constexpr int VEC = 16 / sizeof(TIn);
  double sink = 0.0;
  const int64_t ntiles = (npacks + 256 * U - 1) / (256 * U);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += LOOP ? gridDim.x : ntiles) {
    const int64_t base = tile * (256 * U) + threadIdx.x;
    Pk<TIn, VEC> x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + (int64_t)u * 256;
      if (i < npacks) {
        x[u] = ld<TIn, VEC>(a + VEC * i);
        if (NIN == 2) y[u] = ld<TIn, VEC>(b + VEC * i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + (int64_t)u * 256;
      if (i < npacks) {
        double r[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] = (double)x[u].v[k] + (NIN == 2 ? (double)y[u].v[k] : 0.0);
        if (WRITE) {
#pragma unroll
          for (int k = 0; k < VEC; k += 2) st2<NTS>(out + VEC * i + k, r[k], r[k + 1]);
        } else {
#pragma unroll
          for (int k = 0; k < VEC; ++k) sink += r[k];
        }
      }
    }
  }
  if (!WRITE && sink == 0x1.23456789abcdep+1000) out[0] = sink;
}

// float32 in / float64 out with the access widths of K2's two-column shape: 8-byte loads (2 floats
// per lane), ONE 16-byte store per lane and pack
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NIN, int U, bool LOOP, bool NTS>
__global__ __launch_bounds__(256) void k_probe_f2(const float* __restrict__ a, const float* __restrict__ b,
                                                  int64_t npacks, double* __restrict__ out) {
  const int64_t ntiles = (npacks + 256 * U - 1) / (256 * U);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += LOOP ? gridDim.x : ntiles) {
    const int64_t base = tile * (256 * U) + threadIdx.x;
    f2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + (int64_t)u * 256;
      if (i < npacks) {
        x[u] = __builtin_nontemporal_load(reinterpret_cast<const f2*>(a + 2 * i));
        if (NIN == 2) y[u] = __builtin_nontemporal_load(reinterpret_cast<const f2*>(b + 2 * i));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + (int64_t)u * 256;
      if (i < npacks)
        st2<NTS>(out + 2 * i, (double)x[u].x + (NIN == 2 ? (double)y[u].x : 0.0),
                 (double)x[u].y + (NIN == 2 ? (double)y[u].y : 0.0));
    }
  }
}

static double time_ms(void (*launch)(void*), void* ctx, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(ctx); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0)); launch(ctx); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

struct Ctx { const void* a; const void* b; double* out; int64_t n; int blocks; };

template <typename TIn, int NIN, bool WRITE, int U, bool LOOP, bool NTS>
static void launch(void* p) {
  Ctx* c = (Ctx*)p;
  constexpr int VEC = 16 / sizeof(TIn);
  const int64_t npacks = c->n / VEC;
  const int64_t ntiles = (npacks + 256 * U - 1) / (256 * U);
  const int64_t grid = LOOP ? (c->blocks < ntiles ? c->blocks : ntiles) : ntiles;
  hipLaunchKernelGGL((k_probe<TIn, NIN, WRITE, U, LOOP, NTS>), dim3((unsigned)grid), dim3(256), 0, 0,
                     (const TIn*)c->a, (const TIn*)c->b, npacks, c->out);
}

template <typename TIn, int NIN, bool WRITE, int U>
static void sweep(Ctx c, const char* name) {
  const double bytes = (double)c.n * (NIN * sizeof(TIn) + (WRITE ? 8 : 0));
  const int per_cu[] = {4, 8, 16, 32};
  for (int k = 0; k < 4; ++k) {
    c.blocks = 256 * per_cu[k];
    double ms = time_ms(launch<TIn, NIN, WRITE, U, true, true>, &c, 3);
    printf("%-28s U=%d loop %2d blk/CU nt-store   %8.3f ms %8.1f GB/s\n", name, U, per_cu[k], ms, bytes / ms / 1e6);
  }
  double ms = time_ms(launch<TIn, NIN, WRITE, U, false, true>, &c, 3);
  printf("%-28s U=%d one tile per block nt-store %8.3f ms %8.1f GB/s\n", name, U, ms, bytes / ms / 1e6);
  if (WRITE) {
    ms = time_ms(launch<TIn, NIN, WRITE, U, false, false>, &c, 3);
    printf("%-28s U=%d one tile per block plain-st %8.3f ms %8.1f GB/s\n", name, U, ms, bytes / ms / 1e6);
  }
  fflush(stdout);
}

template <int NIN, int U, bool LOOP, bool NTS>
static void launch_f2(void* p) {
  Ctx* c = (Ctx*)p;
  const int64_t npacks = c->n / 2;
  const int64_t ntiles = (npacks + 256 * U - 1) / (256 * U);
  const int64_t grid = LOOP ? (c->blocks < ntiles ? c->blocks : ntiles) : ntiles;
  hipLaunchKernelGGL((k_probe_f2<NIN, U, LOOP, NTS>), dim3((unsigned)grid), dim3(256), 0, 0,
                     (const float*)c->a, (const float*)c->b, npacks, c->out);
}

template <int NIN, int U>
static void sweep_f2(Ctx c, const char* name) {
  const double bytes = (double)c.n * (NIN * 4 + 8);
  c.blocks = 256 * 32;
  double ms = time_ms(launch_f2<NIN, U, true, true>, &c, 3);
  printf("%-28s U=%d float2 loads, loop 32 blk/CU nt-store %8.3f ms %8.1f GB/s\n", name, U, ms, bytes / ms / 1e6);
  ms = time_ms(launch_f2<NIN, U, false, true>, &c, 3);
  printf("%-28s U=%d float2 loads, one tile per block nt-store %8.3f ms %8.1f GB/s\n", name, U, ms, bytes / ms / 1e6);
  ms = time_ms(launch_f2<NIN, U, false, false>, &c, 3);
  printf("%-28s U=%d float2 loads, one tile per block plain-st %8.3f ms %8.1f GB/s\n", name, U, ms, bytes / ms / 1e6);
  fflush(stdout);
}

template <typename TIn, int NIN, bool WRITE>
static void sweep_u(Ctx c, const char* name) {
  sweep<TIn, NIN, WRITE, 1>(c, name);
  sweep<TIn, NIN, WRITE, 2>(c, name);
  sweep<TIn, NIN, WRITE, 4>(c, name);
  sweep<TIn, NIN, WRITE, 8>(c, name);
}

// fill with values that look like theta/S (a splitmix hash of the index): HBM bandwidth should not
// depend on the data, but power (and with it the clocks) may -- zeros toggle nothing
__global__ void k_fill(double* x, int64_t n, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint64_t z = (uint64_t)i + seed + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31;
    x[i] = -2.0 + 34.0 * (double)(z >> 11) * 0x1.0p-53;
  }
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  const int fill = argc > 2 ? atoi(argv[2]) : 0;      // 1: hashed values instead of zeros
  const int ro = argc > 3 ? atoi(argv[3]) : 0;        // 1: read-only mixes only (no output stream)
  const int64_t n = (int64_t)(gib * (1 << 30) / 8) / 4096 * 4096;  // elements per stream
  void *a, *b; double* out;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc((void**)&out, ro == 1 ? 4096 : n * 8));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(out, 0, ro == 1 ? 4096 : n * 8));
  if (fill) {
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (double*)a, n, 1ull);
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (double*)b, n, 77ull);
    CK(hipDeviceSynchronize());
  }
  printf("elements per stream: %lld (%.1f GiB of float64), %s, %s\n", (long long)n,
         n * 8.0 / (1 << 30), fill ? "hashed values" : "zeros", ro ? "read-only mixes" : "all mixes");
  Ctx c{a, b, out, n, 0};
  if (ro == 2) {  // the float32-in / float64-out mixes only, both load widths
    sweep_u<float, 1, true>(c, "1 x f32 in, 1 x f64 out");
    sweep_f2<1, 1>(c, "1 x f32 in, 1 x f64 out"); sweep_f2<1, 2>(c, "1 x f32 in, 1 x f64 out");
    sweep_f2<1, 4>(c, "1 x f32 in, 1 x f64 out"); sweep_f2<1, 8>(c, "1 x f32 in, 1 x f64 out");
    sweep_u<float, 2, true>(c, "2 x f32 in, 1 x f64 out");
    sweep_f2<2, 1>(c, "2 x f32 in, 1 x f64 out"); sweep_f2<2, 2>(c, "2 x f32 in, 1 x f64 out");
    sweep_f2<2, 4>(c, "2 x f32 in, 1 x f64 out"); sweep_f2<2, 8>(c, "2 x f32 in, 1 x f64 out");
    return 0;
  }
  if (ro) {
    sweep_u<double, 1, false>(c, "1 x f64 in, read-only");
    sweep_u<double, 2, false>(c, "2 x f64 in, read-only");
    return 0;
  }
  sweep_u<double, 1, false>(c, "1 x f64 in, read-only");
  sweep_u<double, 2, false>(c, "2 x f64 in, read-only");
  sweep_u<double, 1, true>(c, "1 x f64 in, 1 x f64 out");
  sweep_u<double, 2, true>(c, "2 x f64 in, 1 x f64 out");
  sweep_u<float, 1, false>(c, "1 x f32 in, read-only");
  sweep_u<float, 2, false>(c, "2 x f32 in, read-only");
  sweep_u<float, 1, true>(c, "1 x f32 in, 1 x f64 out");
  sweep_u<float, 2, true>(c, "2 x f32 in, 1 x f64 out");
  return 0;
}
