// tune_k2.hip -- developer microbenchmark for the K2 (local steric) loop on MI355X.
// Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 scripts/tune_k2.hip -o scripts/tune_k2
//   ./scripts/tune_k2 [nt=16] [rounds=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../momlevel_amd/csrc/eos_device.hpp"
#pragma clang fp contract(off)
using namespace mlx;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
struct D2 {
  double v[2];
};
template <int NTL>
__device__ __forceinline__ D2 ld2(const double* p) {
  f4 raw;
  if constexpr (NTL) raw = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  else raw = *reinterpret_cast<const f4*>(p);
  D2 r;
  __builtin_memcpy(&r, &raw, 16);
  return r;
}
template <int NTS>
__device__ __forceinline__ void st2(double* p, const D2& d) {
  f4 raw;
  __builtin_memcpy(&raw, &d, 16);
  if constexpr (NTS) __builtin_nontemporal_store(raw, reinterpret_cast<f4*>(p));
  else *reinterpret_cast<f4*>(p) = raw;
}

__device__ __forceinline__ double dz_default(double depth, double ztop, double zbot) {
  const double d = is_nan(depth) ? 0.0 : depth;
  const double dz_field = zbot - ztop;
  double part = d - ztop;
  part = (part < 0.0) ? 0.0 : part;
  double result = (part < dz_field) ? part : dz_field;
  part = zbot - 0.0;
  part = (part < 0.0) ? 0.0 : part;
  return (part < result) ? part : result;
}

__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t n) {
  const int64_t q = n / 8, r = n % 8, xcd = b % 8, k = b / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// PF: 0 = loads of a z level issued at the top of its iteration; 1 = next z level's loads issued
// before this level's arithmetic (register double buffer)
template <int NTI, int NTL, int NTS, int PF, int MAP, int DRHO, int BLK = 256>
__global__ __launch_bounds__(BLK) void k2(const double* __restrict__ T, const double* __restrict__ S,
                                          const double* __restrict__ rho0m,
                                          const double* __restrict__ surf,
                                          const double* __restrict__ z_i,
                                          const double* __restrict__ deptho,
                                          const double* __restrict__ p, double c, int nt, int nz,
                                          int64_t plane, int64_t ts, double* __restrict__ drho,
                                          double* __restrict__ eta) {
  const int64_t bx = MAP ? xcd_remap(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  const int64_t col = (bx * BLK + threadIdx.x) * 2;
  if (col + 2 > plane) return;
  const int t0 = blockIdx.y * NTI;
  const int64_t n3 = (int64_t)nz * plane;
  double acc[NTI][2];
#pragma unroll
  for (int j = 0; j < NTI; ++j) acc[j][0] = acc[j][1] = 0.0;
  const D2 depth = ld2<0>(deptho + col);
  D2 a[NTI], b[NTI], na[NTI], nb[NTI];
  auto loads = [&](int z, D2* da, D2* db) {
    const int64_t off = (int64_t)z * plane + col;
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
      const int64_t t = (t0 + j < nt) ? (t0 + j) : (nt - 1);
      da[j] = ld2<NTL>(T + t * ts + off);
      db[j] = ld2<NTL>(S + t * ts + off);
    }
  };
  if constexpr (PF) loads(0, na, nb);
  for (int z = 0; z < nz; ++z) {
    const int64_t off = (int64_t)z * plane + col;
    const D2 r0 = ld2<0>(rho0m + off);
    const double ztop = z_i[z], zbot = z_i[z + 1];
    const double dz0 = dz_default(depth.v[0], ztop, zbot), dz1 = dz_default(depth.v[1], ztop, zbot);
    const double pz = p[z];
    if constexpr (PF) {
#pragma unroll
      for (int j = 0; j < NTI; ++j) {
        a[j] = na[j];
        b[j] = nb[j];
      }
      if (z + 1 < nz) loads(z + 1, na, nb);
    } else {
      loads(z, a, b);
    }
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
      if (t0 + j < nt) {
        D2 d;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const double rho = wright_density<kF64, double>(a[j].v[k], b[j].v[k], pz);
          double dr = rho - r0.v[k];
          dr = is_nan(dr) ? canonical_nan() : dr;
          d.v[k] = dr;
          const double term = (k ? dz1 : dz0) * dr;
          acc[j][k] += is_nan(term) ? 0.0 : term;
        }
        if (DRHO) st2<NTS>(drho + (int64_t)(t0 + j) * n3 + off, d);
      }
    }
  }
  const D2 sf = ld2<0>(surf + col);
#pragma unroll
  for (int j = 0; j < NTI; ++j) {
    if (t0 + j < nt) {
      D2 e;
      e.v[0] = is_nan(sf.v[0]) ? canonical_nan() : c * acc[j][0];
      e.v[1] = is_nan(sf.v[1]) ? canonical_nan() : c * acc[j][1];
      st2<0>(eta + (int64_t)(t0 + j) * plane + col, e);
    }
  }
}

// half-batched: 16 time steps per thread (rho0m amortised over 16) but only 8 steps' loads in
// flight at a time -> fewer VGPRs, 3 waves/SIMD
template <int NTS, int MAP, int DRHO, int PARTS>
__global__ __launch_bounds__(256) void k2hb(const double* __restrict__ T, const double* __restrict__ S,
                                            const double* __restrict__ rho0m,
                                            const double* __restrict__ surf,
                                            const double* __restrict__ z_i,
                                            const double* __restrict__ deptho,
                                            const double* __restrict__ p, double c, int nt, int nz,
                                            int64_t plane, int64_t ts, double* __restrict__ drho,
                                            double* __restrict__ eta) {
  constexpr int NTI = 16, H = NTI / PARTS;
  const int64_t bx = MAP ? xcd_remap(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  const int64_t col = (bx * 256 + threadIdx.x) * 2;
  if (col + 2 > plane) return;
  const int t0 = blockIdx.y * NTI;
  const int64_t n3 = (int64_t)nz * plane;
  double acc[NTI][2];
#pragma unroll
  for (int j = 0; j < NTI; ++j) acc[j][0] = acc[j][1] = 0.0;
  const D2 depth = ld2<0>(deptho + col);
  for (int z = 0; z < nz; ++z) {
    const int64_t off = (int64_t)z * plane + col;
    const D2 r0 = ld2<0>(rho0m + off);
    const double ztop = z_i[z], zbot = z_i[z + 1];
    const double dz0 = dz_default(depth.v[0], ztop, zbot), dz1 = dz_default(depth.v[1], ztop, zbot);
    const double pz = p[z];
#pragma unroll
    for (int h = 0; h < PARTS; ++h) {
      D2 a[H], b[H];
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const int64_t t = (t0 + h * H + j < nt) ? (t0 + h * H + j) : (nt - 1);
        a[j] = ld2<1>(T + t * ts + off);
        b[j] = ld2<1>(S + t * ts + off);
      }
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const int jj = h * H + j;
        if (t0 + jj < nt) {
          D2 d;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const double rho = wright_density<kF64, double>(a[j].v[k], b[j].v[k], pz);
            double dr = rho - r0.v[k];
            dr = is_nan(dr) ? canonical_nan() : dr;
            d.v[k] = dr;
            const double term = (k ? dz1 : dz0) * dr;
            acc[jj][k] += is_nan(term) ? 0.0 : term;
          }
          if (DRHO) st2<NTS>(drho + (int64_t)(t0 + jj) * n3 + off, d);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const D2 sf = ld2<0>(surf + col);
#pragma unroll
  for (int j = 0; j < NTI; ++j) {
    if (t0 + j < nt) {
      D2 e;
      e.v[0] = is_nan(sf.v[0]) ? canonical_nan() : c * acc[j][0];
      e.v[1] = is_nan(sf.v[1]) ? canonical_nan() : c * acc[j][1];
      st2<0>(eta + (int64_t)(t0 + j) * plane + col, e);
    }
  }
}

__global__ void fill(double* x, int64_t n, int64_t n3, double lo, double scale, unsigned long long seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long h = splitmix64(seed ^ (unsigned long long)i);
    const unsigned long long hm = splitmix64(0x1234 ^ (unsigned long long)((i % n3) >> 6));
    const double u = (double)(h >> 11) * 0x1.0p-53;
    x[i] = ((hm & 7) < 2) ? canonical_nan() : lo + scale * u;
  }
}

struct Args {
  const double *T, *S, *rho0m, *surf, *z_i, *deptho, *p;
  int nt, nz;
  int64_t plane, ts;
  double *drho, *eta;
};
struct Variant {
  const char* name;
  void (*launch)(const Args&);
  int bytes_per_cell;
};
template <int NTI, int NTL, int NTS, int PF, int MAP, int DRHO, int BLK = 256>
void launch(const Args& a) {
  dim3 grid((unsigned)((a.plane + 2 * BLK - 1) / (2 * BLK)), (unsigned)((a.nt + NTI - 1) / NTI));
  hipLaunchKernelGGL((k2<NTI, NTL, NTS, PF, MAP, DRHO, BLK>), grid, dim3(BLK), 0, 0, a.T, a.S, a.rho0m,
                     a.surf, a.z_i, a.deptho, a.p, -1.0 / 1035.0, a.nt, a.nz, a.plane, a.ts, a.drho,
                     a.eta);
}

template <int NTS, int MAP, int DRHO, int PARTS>
void launch_hb(const Args& a) {
  dim3 grid((unsigned)((a.plane + 511) / 512), (unsigned)((a.nt + 15) / 16));
  hipLaunchKernelGGL((k2hb<NTS, MAP, DRHO, PARTS>), grid, dim3(256), 0, 0, a.T, a.S, a.rho0m, a.surf,
                     a.z_i, a.deptho, a.p, -1.0 / 1035.0, a.nt, a.nz, a.plane, a.ts, a.drho, a.eta);
}

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 16;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const int nz = 75;
  const int64_t plane = 1080LL * 1440, n3 = plane * nz, n4 = n3 * nt;
  double *T, *S, *rho0m, *dep, *zi, *p, *drho, *eta;
  CK(hipMalloc(&T, n4 * 8));
  CK(hipMalloc(&S, n4 * 8));
  CK(hipMalloc(&drho, n4 * 8));
  CK(hipMalloc(&rho0m, n3 * 8));
  CK(hipMalloc(&dep, plane * 8));
  CK(hipMalloc(&eta, plane * nt * 8));
  CK(hipMalloc(&zi, (nz + 1) * 8));
  CK(hipMalloc(&p, nz * 8));
  hipLaunchKernelGGL(fill, dim3(16384), dim3(256), 0, 0, T, n4, n3, -2.0, 34.0, 1ULL);
  hipLaunchKernelGGL(fill, dim3(16384), dim3(256), 0, 0, S, n4, n3, 30.0, 10.0, 2ULL);
  hipLaunchKernelGGL(fill, dim3(16384), dim3(256), 0, 0, rho0m, n3, n3, 1020.0, 30.0, 3ULL);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, dep, plane, plane, 10.0, 6000.0, 4ULL);
  std::vector<double> ph(nz), zh(nz + 1);
  double zc = 0;
  zh[0] = 0;
  for (int k = 0; k < nz; ++k) {
    const double dz = 2.0 * pow(1.075, k);
    ph[k] = (zc + 0.5 * dz) * 1e4 + 101325.0;
    zc += dz;
    zh[k + 1] = zc;
  }
  CK(hipMemcpy(p, ph.data(), nz * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(zi, zh.data(), (nz + 1) * 8, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  Args a{T, S, rho0m, rho0m, zi, dep, p, nt, nz, plane, n3, drho, eta};

  std::vector<Variant> vs = {
      {"drho NTI16 nt nts xcd B256 ", launch<16, 1, 1, 0, 1, 1, 256>, 24},
      {"drho NTI16 2 half-batches  ", launch_hb<1, 1, 1, 2>, 24},
      {"drho NTI16 4 quarter-batch ", launch_hb<1, 1, 1, 4>, 24},
      {"eta  NTI16 nt     xcd B256 ", launch<16, 1, 0, 0, 1, 0, 256>, 16},
      {"eta  NTI16 2 half-batches  ", launch_hb<1, 1, 0, 2>, 16},
      {"eta  NTI16 4 quarter-batch ", launch_hb<1, 1, 0, 4>, 16},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<std::vector<float>> ms(vs.size());
  for (int r = 0; r < rounds + 1; ++r)
    for (size_t i = 0; i < vs.size(); ++i) {
      CK(hipEventRecord(e0, 0));
      vs[i].launch(a);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[i].push_back(t);
    }
  printf("nt=%d  cells=%.3e\n", nt, (double)n4);
  for (size_t i = 0; i < vs.size(); ++i) {
    std::sort(ms[i].begin(), ms[i].end());
    const float mn = ms[i].front(), md = ms[i][ms[i].size() / 2];
    const double bytes = (double)vs[i].bytes_per_cell * n4;
    printf("%-30s min %8.3f ms (%6.0f GB/s alg, %5.1f%%)  median %8.3f ms  %7.0f Mcells/s\n",
           vs[i].name, mn, bytes / mn / 1e6, bytes / mn / 1e6 / 80.0, md, n4 / md / 1e3);
  }
  return 0;
}
