// tune_streams.hip -- where do K1's last few percent go?  Pure streaming kernels, 16-byte nt loads:
//   A  one stream, grid-stride (mlx_nansum's pattern: the "read probe")
//   B  two streams (theta and S, separate allocations), grid-stride
//   C  two streams, K1's tiling and time loop (thread = 4 packs of one z level, 32-step chunks,
//      register double buffering), one add per cell
//   D  as C, but every block walks its tile through ALL time steps of the chunk list in a
//      persistent grid (256 x 8 blocks), tiles dealt contiguously per XCD
//   hipcc --offload-arch=gfx950 -O3 scripts/tune_streams.hip -o scripts/tune_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP %s at %d\n",hipGetErrorString(e_),__LINE__);exit(1);} } while(0)
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 ld(const double* p) {
  f4 r = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  double2 d; __builtin_memcpy(&d, &r, 16); return d;
}
__device__ __forceinline__ double wsum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void kA(const double* x, int64_t n2, double* out) {
  double c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    double2 v = ld(x + 2 * i); c += v.x + v.y;
  }
  c = wsum(c); if ((threadIdx.x & 63) == 0) atomicAdd(out, c);
}
__global__ __launch_bounds__(256) void kB(const double* x, const double* y, int64_t n2, double* out) {
  double c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    double2 v = ld(x + 2 * i), w = ld(y + 2 * i); c += (v.x + v.y) + (w.x + w.y);
  }
  c = wsum(c); if ((threadIdx.x & 63) == 0) atomicAdd(out, c);
}
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t n) {
  const int64_t q = n / 8, r = n % 8, xcd = b % 8, k = b / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
// K1-like: grid (tiles, nz, chunks)
template <int U>
__global__ __launch_bounds__(256) void kC(const double* T, const double* S, int nt, int tchunk, int64_t plane,
                                          int64_t n3, double* out) {
  const int64_t bx = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t tile0 = bx * (256 * 2 * U);
  const int64_t zoff = (int64_t)blockIdx.y * plane;
  const int tb = blockIdx.z * tchunk, te = min(tb + tchunk, nt);
  int64_t off[U];
  for (int u = 0; u < U; ++u) { int64_t i = tile0 + ((int64_t)u * 256 + threadIdx.x) * 2; off[u] = zoff + (i + 2 <= plane ? i : 0); }
  double2 nT[U], nS[U], cT[U], cS[U];
  for (int u = 0; u < U; ++u) { nT[u] = ld(T + (int64_t)tb * n3 + off[u]); nS[u] = ld(S + (int64_t)tb * n3 + off[u]); }
  double c = 0;
  for (int t = tb; t < te; ++t) {
    for (int u = 0; u < U; ++u) { cT[u] = nT[u]; cS[u] = nS[u]; }
    if (t + 1 < te) for (int u = 0; u < U; ++u) { nT[u] = ld(T + (int64_t)(t + 1) * n3 + off[u]); nS[u] = ld(S + (int64_t)(t + 1) * n3 + off[u]); }
    for (int u = 0; u < U; ++u) c += (cT[u].x + cT[u].y) + (cS[u].x + cS[u].y);
  }
  c = wsum(c); if ((threadIdx.x & 63) == 0) atomicAdd(out, c);
}
// persistent: nblk blocks; block b owns tiles b, b+nblk, ... of the (z,tile) list; for each tile
// walks all time steps of chunk after chunk
template <int U>
__global__ __launch_bounds__(256) void kD(const double* T, const double* S, int nt, int tchunk, int64_t plane,
                                          int64_t n3, int64_t tiles_per_z, int nz, double* out) {
  const int64_t ntiles = tiles_per_z * nz;
  double c = 0;
  for (int tb = 0; tb < nt; tb += tchunk) {
    const int te = min(tb + tchunk, nt);
    for (int64_t w = xcd_remap(blockIdx.x, gridDim.x); w < ntiles; w += gridDim.x) {
      const int64_t z = w / tiles_per_z, bx = w % tiles_per_z;
      const int64_t tile0 = bx * (256 * 2 * U), zoff = z * plane;
      int64_t off[U];
      for (int u = 0; u < U; ++u) { int64_t i = tile0 + ((int64_t)u * 256 + threadIdx.x) * 2; off[u] = zoff + (i + 2 <= plane ? i : 0); }
      double2 nT[U], nS[U], cT[U], cS[U];
      for (int u = 0; u < U; ++u) { nT[u] = ld(T + (int64_t)tb * n3 + off[u]); nS[u] = ld(S + (int64_t)tb * n3 + off[u]); }
      for (int t = tb; t < te; ++t) {
        for (int u = 0; u < U; ++u) { cT[u] = nT[u]; cS[u] = nS[u]; }
        if (t + 1 < te) for (int u = 0; u < U; ++u) { nT[u] = ld(T + (int64_t)(t + 1) * n3 + off[u]); nS[u] = ld(S + (int64_t)(t + 1) * n3 + off[u]); }
        for (int u = 0; u < U; ++u) c += (cT[u].x + cT[u].y) + (cS[u].x + cS[u].y);
      }
    }
  }
  c = wsum(c); if ((threadIdx.x & 63) == 0) atomicAdd(out, c);
}
// time-major: all resident blocks sweep ONE time step's slab together (tiles fastest), then the next
template <int U>
__global__ __launch_bounds__(256) void kE(const double* T, const double* S, int nt, int64_t n3, double* out) {
  // grid-stride over (t, pack) with t slowest: a moving front through T[t] and S[t]
  const int64_t packs = n3 / 2;
  double c = 0;
  for (int t = 0; t < nt; ++t) {
    const double* Tt = T + (int64_t)t * n3; const double* St = S + (int64_t)t * n3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < packs; i += (int64_t)gridDim.x * 256) {
      double2 v = ld(Tt + 2 * i), w = ld(St + 2 * i); c += (v.x + v.y) + (w.x + w.y);
    }
  }
  c = wsum(c); if ((threadIdx.x & 63) == 0) atomicAdd(out, c);
}
__global__ void kfill(double* x, int64_t n, unsigned long long seed) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    unsigned long long z = seed + (unsigned long long)i * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    x[i] = -2.0 + 34.0 * ((double)(z >> 11) * 0x1.0p-53);
  }
}
int main(int argc, char** argv) {
  const bool zeros = argc > 1 && argv[1][0] == 'z';
  const int nt = 56, nz = 75; const int64_t plane = 1080 * 1440, n3 = nz * plane, n = (int64_t)nt * n3;
  double *T, *S, *out;
  CK(hipMalloc(&T, n * 8)); CK(hipMalloc(&S, n * 8)); CK(hipMalloc(&out, 8));
  CK(hipMemset(out, 0, 8));
  if (zeros) { CK(hipMemset(T, 0, n * 8)); CK(hipMemset(S, 0, n * 8)); }
  else { hipLaunchKernelGGL(kfill, dim3(16384), dim3(256), 0, 0, T, n, 1ULL); hipLaunchKernelGGL(kfill, dim3(16384), dim3(256), 0, 0, S, n, 2ULL); CK(hipDeviceSynchronize()); }
  printf("# data: %s\n", zeros ? "all zeros" : "random doubles");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, double bytes, auto launch) {
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
    }
    printf("%-58s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6); fflush(stdout);
  };
  const double b1 = (double)n * 8, b2 = 2.0 * n * 8;
  for (int nb : {2048, 8192, 32768})
    run((std::string("A one stream, grid-stride, blocks=") + std::to_string(nb)).c_str(), b1, [&] { hipLaunchKernelGGL(kA, dim3(nb), dim3(256), 0, 0, T, n / 2, out); });
  for (int nb : {2048, 8192, 32768})
    run((std::string("B two streams, grid-stride, blocks=") + std::to_string(nb)).c_str(), b2, [&] { hipLaunchKernelGGL(kB, dim3(nb), dim3(256), 0, 0, T, S, n / 2, out); });
  for (int nb : {2048, 8192})
    run((std::string("E two streams, time-major sweep, blocks=") + std::to_string(nb)).c_str(), b2, [&] { hipLaunchKernelGGL(kE<1>, dim3(nb), dim3(256), 0, 0, T, S, nt, n3, out); });
  for (int tc : {8, 28, 56}) {
    run((std::string("C K1 tiling U=4, t_chunk=") + std::to_string(tc)).c_str(), b2, [&] {
      hipLaunchKernelGGL(kC<4>, dim3((plane + 2047) / 2048, nz, (nt + tc - 1) / tc), dim3(256), 0, 0, T, S, nt, tc, plane, n3, out); });
    run((std::string("C K1 tiling U=2, t_chunk=") + std::to_string(tc)).c_str(), b2, [&] {
      hipLaunchKernelGGL(kC<2>, dim3((plane + 1023) / 1024, nz, (nt + tc - 1) / tc), dim3(256), 0, 0, T, S, nt, tc, plane, n3, out); });
  }
  for (int nb : {1024, 2048, 4096})
    for (int tc : {8, 28})
      run((std::string("D persistent U=4, blocks=") + std::to_string(nb) + " t_chunk=" + std::to_string(tc)).c_str(), b2, [&] {
        hipLaunchKernelGGL(kD<4>, dim3(nb), dim3(256), 0, 0, T, S, nt, tc, plane, n3, (plane + 2047) / 2048, nz, out); });
  return 0;
}
