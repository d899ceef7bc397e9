// tune_k1.hip -- developer microbenchmark for the K1 (global steric) inner loop on MI355X.
// Not part of the product.  Build (cross-compiles here) and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 scripts/tune_k1.hip -o scripts/tune_k1
//   ./scripts/tune_k1 [nt=16] [rounds=5]
// Variants are timed interleaved in ONE process (cdna_hip_programming.md rule 24).
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

// reciprocal without the div_scale/div_fixup range handling: same Newton sequence as the IEEE
// expansion, identical result whenever no scaling is needed (|x| in a sane range)
__device__ __forceinline__ double rcp_noscale(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  // final correction step: q = r (numerator 1), rem = fma(-x, q, 1), q + rem*r
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(e, r, r);
}

template <int MATH>
__device__ __forceinline__ double rho_of(double T, double S, double p) {
  if constexpr (MATH == 2) {
    return T + S;
  } else {
    double al0, p0, lam;
    wright_terms<double>(T, S, al0, p0, lam);
    const double pp0 = p + p0;
    const double den = lam + al0 * pp0;
    double I;
    if constexpr (MATH == 1) I = rcp_noscale(den);
    else I = 1.0 / den;
    return pp0 * I;
  }
}

constexpr int NTC = 8;

// MAP: 0 = blockIdx.x linear; 1 = XCD-contiguous remap (blocks that share an XCD walk one
// contiguous eighth of the plane)
template <int BLK, int U, int PF, int NTL, int MATH, int MAP>
__global__ __launch_bounds__(BLK) void k1(const double* __restrict__ T, const double* __restrict__ S,
                                          const double* __restrict__ vol0,
                                          const double* __restrict__ p, int nt, int64_t plane,
                                          int64_t ts, double* __restrict__ partials,
                                          int64_t nblk_total) {
  __shared__ double red[NTC][BLK];
  const int tid = threadIdx.x;
  int z = blockIdx.y;
  int64_t bx = blockIdx.x;
  if constexpr (MAP == 1) {
    const int64_t n = gridDim.x, q = n / 8, r = n % 8, xcd = bx % 8, k = bx / 8;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  if constexpr (MAP == 2) {  // remap over the whole (x,z) id space
    const int64_t id = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int64_t n = (int64_t)gridDim.x * gridDim.y, q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
    const int64_t nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    z = (int)(nid / gridDim.x);
    bx = nid % gridDim.x;
  }
  const int64_t blk = (int64_t)z * gridDim.x + bx;
  const int64_t tile0 = bx * (int64_t)(BLK * 2 * U);
  const int64_t zoff = (int64_t)z * plane;
  int64_t off[U];
  double vol[U][2];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int64_t i = tile0 + ((int64_t)u * BLK + tid) * 2;
    const bool valid = (i + 2 <= plane);
    i = valid ? i : 0;
    off[u] = zoff + i;
    D2 v = ld2<0>(vol0 + off[u]);
    vol[u][0] = valid ? v.v[0] : canonical_nan();
    vol[u][1] = valid ? v.v[1] : canonical_nan();
  }
  const double pz = p[z];
  D2 cT[U], cS[U], nT[U], nS[U];
  if constexpr (PF) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      nT[u] = ld2<NTL>(T + off[u]);
      nS[u] = ld2<NTL>(S + off[u]);
    }
  }
  for (int t = 0; t < nt; ++t) {
    if constexpr (PF) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        cT[u] = nT[u];
        cS[u] = nS[u];
      }
      if (t + 1 < nt) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          nT[u] = ld2<NTL>(T + (int64_t)(t + 1) * ts + off[u]);
          nS[u] = ld2<NTL>(S + (int64_t)(t + 1) * ts + off[u]);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        cT[u] = ld2<NTL>(T + (int64_t)t * ts + off[u]);
        cS[u] = ld2<NTL>(S + (int64_t)t * ts + off[u]);
      }
    }
    double c = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double term = rho_of<MATH>(cT[u].v[k], cS[u].v[k], pz) * vol[u][k];
        c += is_nan(term) ? 0.0 : term;
      }
    const int row = t % NTC;
    red[row][tid] = c;
    if (row == NTC - 1 || t == nt - 1) {
      __syncthreads();
      const int wave = tid >> 6, lane = tid & 63;
      for (int r = wave; r <= row; r += BLK / 64) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) v += red[r][lane + 64 * w];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) partials[(int64_t)(t - row + r) * nblk_total + blk] = v;
      }
      __syncthreads();
    }
  }
}

// buffer_load variant: per-block SRSRC descriptor (base = uniform), 32-bit voffset, AUX cache bits
// (1 = sc0, 2 = nt, 16 = sc1).  Out-of-range lanes read 0 (hardware bounds check).
template <int U, int AUX, int MATH>
__global__ __launch_bounds__(256) void k1buf(const double* __restrict__ T, const double* __restrict__ S,
                                             const double* __restrict__ vol0,
                                             const double* __restrict__ p, int nt, int64_t plane,
                                             int64_t ts, double* __restrict__ partials,
                                             int64_t nblk_total) {
  constexpr int BLK = 256;
  __shared__ double red[NTC][BLK];
  const int tid = threadIdx.x;
  const int z = blockIdx.y;
  const int64_t bx = blockIdx.x;
  const int64_t blk = (int64_t)blockIdx.y * gridDim.x + bx;
  const int64_t tile0 = bx * (int64_t)(BLK * 2 * U);
  const int64_t zoff = (int64_t)z * plane;
  const int64_t rem = plane - tile0;
  const unsigned nbytes = (unsigned)((rem < (int64_t)BLK * 2 * U ? rem : (int64_t)BLK * 2 * U) * 8);
  double vol[U][2];
  unsigned voff[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    voff[u] = (unsigned)((u * BLK + tid) * 16);
    const bool valid = voff[u] + 16 <= nbytes;
    D2 v = ld2<0>(vol0 + zoff + tile0 + (valid ? voff[u] / 8 : 0));
    vol[u][0] = valid ? v.v[0] : canonical_nan();
    vol[u][1] = valid ? v.v[1] : canonical_nan();
  }
  const double pz = p[z];
  const double* bT = T + zoff + tile0;
  const double* bS = S + zoff + tile0;
  D2 cT[U], cS[U], nT[U], nS[U];
  auto loadall = [&](const double* baseT, const double* baseS, D2* dT, D2* dS) {
    auto rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(baseT), 0, nbytes, 0x00020000);
    auto rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(baseS), 0, nbytes, 0x00020000);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      u4 a = __builtin_amdgcn_raw_buffer_load_b128(rT, voff[u], 0, AUX);
      u4 b = __builtin_amdgcn_raw_buffer_load_b128(rS, voff[u], 0, AUX);
      __builtin_memcpy(&dT[u], &a, 16);
      __builtin_memcpy(&dS[u], &b, 16);
    }
  };
  loadall(bT, bS, nT, nS);
  for (int t = 0; t < nt; ++t) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cT[u] = nT[u];
      cS[u] = nS[u];
    }
    if (t + 1 < nt) loadall(bT + (int64_t)(t + 1) * ts, bS + (int64_t)(t + 1) * ts, nT, nS);
    double c = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double term = rho_of<MATH>(cT[u].v[k], cS[u].v[k], pz) * vol[u][k];
        c += is_nan(term) ? 0.0 : term;
      }
    const int row = t % NTC;
    red[row][tid] = c;
    if (row == NTC - 1 || t == nt - 1) {
      __syncthreads();
      const int wave = tid >> 6, lane = tid & 63;
      for (int r = wave; r <= row; r += BLK / 64) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) v += red[r][lane + 64 * w];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) partials[(int64_t)(t - row + r) * nblk_total + blk] = v;
      }
      __syncthreads();
    }
  }
}

template <int U, int AUX, int MATH>
void launchbuf(const double* T, const double* S, const double* v, const double* p, int nt,
               int64_t plane, int64_t ts, double* partials, int nz, hipStream_t st) {
  const int64_t gx = (plane + (int64_t)256 * 2 * U - 1) / ((int64_t)256 * 2 * U);
  hipLaunchKernelGGL((k1buf<U, AUX, MATH>), dim3((unsigned)gx, nz), dim3(256), 0, st, T, S, v, p,
                     nt, plane, ts, partials, gx * nz);
}

// one launch, time chunk as the slowest grid dimension: block (x, z, c) handles steps [c*CH, c*CH+CH)
template <int BLK, int U, int NTL, int MAP>
__global__ __launch_bounds__(BLK) void k1z(const double* __restrict__ T, const double* __restrict__ S,
                                           const double* __restrict__ vol0,
                                           const double* __restrict__ p, int nt, int ch, int64_t plane,
                                           int64_t ts, double* __restrict__ partials,
                                           int64_t nblk_total) {
  __shared__ double red[NTC][BLK];
  const int tid = threadIdx.x;
  const int z = blockIdx.y;
  const int tb = blockIdx.z * ch;
  const int te = (tb + ch < nt) ? (tb + ch) : nt;
  int64_t bx = blockIdx.x;
  if constexpr (MAP == 1) {
    const int64_t n = gridDim.x, q = n / 8, r = n % 8, xcd = bx % 8, k = bx / 8;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int64_t blk = (int64_t)z * gridDim.x + bx;
  const int64_t tile0 = bx * (int64_t)(BLK * 2 * U);
  const int64_t zoff = (int64_t)z * plane;
  int64_t off[U];
  double vol[U][2];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int64_t i = tile0 + ((int64_t)u * BLK + tid) * 2;
    const bool valid = (i + 2 <= plane);
    i = valid ? i : 0;
    off[u] = zoff + i;
    D2 v = ld2<0>(vol0 + off[u]);
    vol[u][0] = valid ? v.v[0] : canonical_nan();
    vol[u][1] = valid ? v.v[1] : canonical_nan();
  }
  const double pz = p[z];
  D2 cT[U], cS[U], nT[U], nS[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    nT[u] = ld2<NTL>(T + (int64_t)tb * ts + off[u]);
    nS[u] = ld2<NTL>(S + (int64_t)tb * ts + off[u]);
  }
  for (int t = tb; t < te; ++t) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cT[u] = nT[u];
      cS[u] = nS[u];
    }
    if (t + 1 < te) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        nT[u] = ld2<NTL>(T + (int64_t)(t + 1) * ts + off[u]);
        nS[u] = ld2<NTL>(S + (int64_t)(t + 1) * ts + off[u]);
      }
    }
    double c = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double term = rho_of<0>(cT[u].v[k], cS[u].v[k], pz) * vol[u][k];
        c += is_nan(term) ? 0.0 : term;
      }
    const int row = (t - tb) % NTC;
    red[row][tid] = c;
    if (row == NTC - 1 || t == te - 1) {
      __syncthreads();
      const int wave = tid >> 6, lane = tid & 63;
      for (int r = wave; r <= row; r += BLK / 64) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) v += red[r][lane + 64 * w];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) partials[(int64_t)(t - row + r) * nblk_total + blk] = v;
      }
      __syncthreads();
    }
  }
}

template <int BLK, int U, int NTL, int MAP, int CH>
void launch_z(const double* T, const double* S, const double* v, const double* p, int nt,
              int64_t plane, int64_t ts, double* partials, int nz, hipStream_t st) {
  const int64_t gx = (plane + (int64_t)BLK * 2 * U - 1) / ((int64_t)BLK * 2 * U);
  hipLaunchKernelGGL((k1z<BLK, U, NTL, MAP>), dim3((unsigned)gx, nz, (nt + CH - 1) / CH), dim3(BLK),
                     0, st, T, S, v, p, nt, CH, plane, ts, partials, gx * nz);
}

template <int BLK, int U, int PF, int NTL, int MATH, int MAP, int CH>
void launch_split(const double* T, const double* S, const double* v, const double* p, int nt,
                  int64_t plane, int64_t ts, double* partials, int nz, hipStream_t st) {
  const int64_t gx = (plane + (int64_t)BLK * 2 * U - 1) / ((int64_t)BLK * 2 * U);
  for (int t0 = 0; t0 < nt; t0 += CH) {
    const int n = (nt - t0 < CH) ? (nt - t0) : CH;
    hipLaunchKernelGGL((k1<BLK, U, PF, NTL, MATH, MAP>), dim3((unsigned)gx, nz), dim3(BLK), 0, st,
                       T + (int64_t)t0 * ts, S + (int64_t)t0 * ts, v, p, n, plane, ts,
                       partials + (int64_t)t0 * gx * nz, gx * nz);
  }
}

__global__ void fill(double* x, int64_t n, int64_t n3, double lo, double scale, unsigned long long seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long h = splitmix64(seed ^ (unsigned long long)i);
    const unsigned long long hm = splitmix64(0x1234 ^ (unsigned long long)((i % n3) >> 6));
    const double u = (double)(h >> 11) * 0x1.0p-53;
    x[i] = ((hm & 7) < 2) ? canonical_nan() : lo + scale * u;  // ~25% NaN in 64-cell runs
  }
}

struct Variant {
  const char* name;
  void (*launch)(const double*, const double*, const double*, const double*, int, int64_t, int64_t,
                 double*, int nz, hipStream_t);
};

template <int BLK, int U, int PF, int NTL, int MATH, int MAP>
void launch(const double* T, const double* S, const double* v, const double* p, int nt,
            int64_t plane, int64_t ts, double* partials, int nz, hipStream_t st) {
  const int64_t gx = (plane + (int64_t)BLK * 2 * U - 1) / ((int64_t)BLK * 2 * U);
  hipLaunchKernelGGL((k1<BLK, U, PF, NTL, MATH, MAP>), dim3((unsigned)gx, nz), dim3(BLK), 0, st, T,
                     S, v, p, nt, plane, ts, partials, gx * nz);
}

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 16;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const int nz = 75;
  const int64_t plane = 1080LL * 1440, n3 = plane * nz, n4 = n3 * nt;
  double *T, *S, *vol, *p, *partials;
  const int nt_alloc = getenv("NTALLOC") ? atoi(getenv("NTALLOC")) : nt;
  const int64_t n4a = n3 * nt_alloc;
  CK(hipMalloc(&T, n4a * 8));
  const int64_t spad = getenv("SPAD") ? atoll(getenv("SPAD")) : 0;  // bytes of skew between T and S
  CK(hipMalloc(&S, n4a * 8 + spad));
  S = (double*)((char*)S + spad);
  CK(hipMalloc(&vol, n3 * 8));
  CK(hipMalloc(&p, nz * 8));
  CK(hipMalloc(&partials, (size_t)nt * 2 * (n3 / 512 + 1024) * 8));
  hipLaunchKernelGGL(fill, dim3(16384), dim3(256), 0, 0, T, n4a, n3, -2.0, 34.0, 1ULL);
  hipLaunchKernelGGL(fill, dim3(16384), dim3(256), 0, 0, S, n4a, n3, 30.0, 10.0, 2ULL);
  hipLaunchKernelGGL(fill, dim3(16384), dim3(256), 0, 0, vol, n3, n3, 1e9, 1e11, 3ULL);
  std::vector<double> ph(nz);
  double zc = 0;
  for (int k = 0; k < nz; ++k) {
    const double dz = 2.0 * pow(1.075, k);
    ph[k] = (zc + 0.5 * dz) * 1e4 + 101325.0;
    zc += dz;
  }
  CK(hipMemcpy(p, ph.data(), nz * 8, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());

  std::vector<Variant> vs = {
      {"nt xcd1 B256 U4 zchunk 24     ", launch_z<256, 4, 1, 1, 24>},
      {"nt xcd1 B256 U4 zchunk 32     ", launch_z<256, 4, 1, 1, 32>},
      {"nt xcd1 B256 U4 zchunk 40     ", launch_z<256, 4, 1, 1, 40>},
      {"nt xcd1 B256 U4 zchunk 48     ", launch_z<256, 4, 1, 1, 48>},
      {"nt xcd1 B256 U4 zchunk 64     ", launch_z<256, 4, 1, 1, 64>},
      {"nt xcd1 B256 U4 one launch    ", launch<256, 4, 1, 1, 0, 1>},
  };
  printf("T=%p S=%p spad=%lld\n", (void*)T, (void*)S, (long long)spad);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<std::vector<float>> ms(vs.size());
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t i = 0; i < vs.size(); ++i) {
      CK(hipEventRecord(e0, 0));
      vs[i].launch(T, S, vol, p, nt, plane, getenv("TS0") ? (int64_t)atoll(getenv("TS0")) * n3 / 1000 : n3, partials, nz, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[i].push_back(t);
    }
  }
  const double bytes = 16.0 * n4;
  printf("nt=%d  %.1f GB streamed per launch\n", nt, bytes / 1e9);
  for (size_t i = 0; i < vs.size(); ++i) {
    printf("  in order:");
    for (float x : ms[i]) printf(" %.2f", x);
    printf("\n");
    std::sort(ms[i].begin(), ms[i].end());
    const float mn = ms[i].front(), md = ms[i][ms[i].size() / 2];
    printf("%-32s min %8.3f ms (%6.0f GB/s, %5.1f%% of 8TB/s)  median %8.3f ms (%6.0f GB/s)\n",
           vs[i].name, mn, bytes / mn / 1e6, bytes / mn / 1e6 / 80.0, md, bytes / md / 1e6);
  }
  return 0;
}
