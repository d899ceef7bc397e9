// momlevel_hip.hip -- HIP kernels (gfx950 / CDNA4) and the C ABI of libmomlevel_hip.so.
//
// Hot path: momlevel's steric sea level (src/momlevel/steric.py:17-184) =
//   Wright EOS per cell (eos/wright.py:44-48)
//   + global:  sum_{z,y,x} rho*vol0 per time step   (derived.py:435-438)
//   + local:   delta_rho = rho - rho0, eta = -1/rhozero * sum_z dz*delta_rho (steric.py:151-166)
//
// Everything is pointwise + reduction: ~50 fp64 VALU ops against 16 B of HBM
// traffic per cell.  The bound is HBM bandwidth, there is no contraction and
// therefore no MFMA.  The design rules that matter (cdna_hip_programming.md G2, G7,
// G11, G13; Appendix B element-wise/reduction):
//   * 16 B per lane loads (global_load_dwordx4): double2 / float4, 64 lanes = 1 KiB
//     contiguous per wave instruction;
//   * time is the INNER loop of every thread, so that the time-invariant operands
//     (vol0, rho0, dz, p) are read once and live in VGPRs while theta/S stream by;
//   * next-time-step loads are issued before the current step's arithmetic
//     (register double buffering) so each wave keeps 2x its working set in flight;
//   * reductions are wave/LDS trees in a fixed order -- no float atomics -- so results
//     are bit-reproducible run to run and independent of dispatch order;
//   * >> 256 workgroups per launch (tens of thousands), each independent: no
//     inter-workgroup hand-off inside a launch, so nothing depends on XCD placement.
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared  (csrc/build.py)

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <initializer_list>
#include <type_traits>

#include "../../include/momlevel_hip.h"
#include "eos_device.hpp"
#include "mlx_internal.hpp"

#pragma clang fp contract(off)

namespace mlx {

constexpr int kBlock = 256;  // 4 waves of 64
constexpr int kNTC = 8;      // time steps whose per-thread partials are parked in LDS

// ------------------------------------------------------------------------------------
// vector load helpers: VEC elements of TIn = 16 bytes (double2 / float4), or scalar
// ------------------------------------------------------------------------------------
template <typename TIn, int VEC>
struct Pack {
  TIn v[VEC];
};

typedef float f4_t __attribute__((ext_vector_type(4)));

// STREAM: once-read data (theta/S, delta_rho) moves with the `nt` cache policy
// (global_load_dwordx4 ... nt).  Measured on MI355X (scripts/tune_k1.hip): the K1 loop
// streams at ~6.0 TB/s with default-policy loads and ~6.6 TB/s with nt loads.
template <bool STREAM>
__device__ __forceinline__ f4_t load16(const void* p) {
  if constexpr (STREAM) return __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
  else return *reinterpret_cast<const f4_t*>(p);
}

template <typename TIn, int VEC, bool STREAM = false>
__device__ __forceinline__ Pack<TIn, VEC> load_pack(const TIn* __restrict__ p) {
  Pack<TIn, VEC> r;
  if constexpr (VEC == 1) {
    if constexpr (STREAM) r.v[0] = __builtin_nontemporal_load(p);
    else r.v[0] = p[0];
  } else if constexpr (sizeof(TIn) * VEC == 8) {  // float2: one global_load_dwordx2
    double raw;
    if constexpr (STREAM) raw = __builtin_nontemporal_load(reinterpret_cast<const double*>(p));
    else raw = *reinterpret_cast<const double*>(p);
    __builtin_memcpy(&r, &raw, 8);
  } else if constexpr (sizeof(TIn) * VEC == 16) {
    f4_t raw = load16<STREAM>(p);  // one global_load_dwordx4
    __builtin_memcpy(&r, &raw, 16);
  } else {
    static_assert(sizeof(TIn) * VEC == 32, "pack must be 16 or 32 bytes");
    f4_t raw0 = load16<STREAM>(p);
    f4_t raw1 = load16<STREAM>(reinterpret_cast<const char*>(p) + 16);
    __builtin_memcpy(&r.v[0], &raw0, 16);
    __builtin_memcpy(&r.v[VEC / 2], &raw1, 16);
  }
  return r;
}

// theta / salinity loads at element offset `off`.  In the mixed-dtype modes (generic kernels only,
// TIn = double) ONE of the two fields is float32 in memory: it is read as float and widened
// (exact), so the kernel body sees doubles holding float32 values and eos_eval<kMix*> narrows them
// back (exact) for that field's float32 part of the polynomial.
template <int MODE, bool IS_T, typename TIn, int VEC, bool STREAM = false>
__device__ __forceinline__ Pack<TIn, VEC> load_field(const TIn* __restrict__ base, int64_t off) {
  constexpr bool F32_HERE = (MODE == kMixT32 && IS_T) || (MODE == kMixS32 && !IS_T);
  if constexpr (F32_HERE) {
    static_assert((VEC == 1 || VEC == 2) && sizeof(TIn) == 8, "the other field is float64");
    Pack<TIn, VEC> r;
    const float* q = reinterpret_cast<const float*>(base) + off;
    if constexpr (VEC == 2) {  // fast kernels: the two float32 cells of the pack in one 8-byte load
      const Pack<float, 2> f = load_pack<float, 2, STREAM>(q);
      r.v[0] = f.v[0];
      r.v[1] = f.v[1];
    } else if constexpr (STREAM) {
      r.v[0] = __builtin_nontemporal_load(q);
    } else {
      r.v[0] = q[0];
    }
    return r;
  } else {
    return load_pack<TIn, VEC, STREAM>(base + off);
  }
}

// XCD-aware tile index: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share
// one), so give the blocks that share an XCD one contiguous eighth of the row of tiles.
// Bijective for any n (cdna_hip_programming.md T1); speed only, never correctness.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t n) {
  const int64_t q = n / 8, r = n % 8, xcd = b % 8, k = b / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int VEC, bool STREAM = false>
__device__ __forceinline__ void store_pack(double* __restrict__ p, const Pack<double, VEC>& r) {
  if constexpr (VEC == 1) {
    if constexpr (STREAM) __builtin_nontemporal_store(r.v[0], p);
    else p[0] = r.v[0];
  } else {
#pragma unroll
    for (int h = 0; h < VEC / 2; ++h) {
      f4_t raw;
      __builtin_memcpy(&raw, &r.v[2 * h], 16);
      if constexpr (STREAM) __builtin_nontemporal_store(raw, reinterpret_cast<f4_t*>(p) + h);
      else reinterpret_cast<f4_t*>(p)[h] = raw;
    }
  }
}

// 16 bytes per lane HBM -> LDS with no VGPR destination (global_load_lds_dwordx4 ... nt): lane l of
// the wave writes lds_slice + 16*l, lds_slice being the wave's (uniform) 1 KiB slice.  Asynchronous:
// counted by vmcnt; the data is in LDS after the caller's own s_waitcnt.  (The builtin exists in the
// device pass only.)
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_slice) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_slice, 16, 0,
                                   2 /* nt */);
#endif
}

__device__ __forceinline__ double wave_sum(double v) {
  // fixed-order butterfly-free tree: lane i += lane i+off, off = 32..1; lane 0 holds the sum
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------------------
// K1: fused EOS + rho*vol0 + per-time-step sum over the block's cells.
//
// grid = (ceil(plane / (kBlock*VEC*U)), nz, ceil(nt/t_chunk)); each thread owns U packs of
// VEC adjacent cells of ONE z level, keeps their vol0 (and p, if FULL3D) in registers and
// loops over the time steps of its chunk.  Per time step it parks its partial in LDS row
// (t - chunk start) % NTC; every NTC steps the block reduces the parked rows (fixed order) and
// writes partials[t][block].  A second kernel (k_reduce_rows) sums partials[t][:] in a fixed
// order -> masso[t].
//
// VAR (steric.py:115-125): 0 = steric, both fields stream; 1 = halosteric, theta is held at the
// reference state T0; 2 = thermosteric, S is held at S0; 3 = ALL THREE in one pass over theta/S
// (+ the heat-content integrand sum(theta*vol0), an extension) -- 4 partials per step,
// partials[(o*nt + t)][block], o = 0 steric, 1 thermosteric, 2 halosteric, 3 heat.  The held
// field is read once and its part of the polynomial (eos_device.hpp TPart/SPart) is evaluated
// once, outside the time loop; VAR 3 also shares the streamed fields' parts between the
// variants.  Every variant adds the same terms in the same order from the same tiling, so each
// output of VAR 3 is bit-identical to the corresponding single-variant launch, and masso(t=0)
// of all of them to the reference state's masso0.
// GENERIC (VEC==1 instantiation) takes eos/p_mode at run time, incl. a time-dependent
// pressure (MLX_P_FULL4D), and evaluates every density from scratch.
// SKIP (MLX_FLAG_SKIP_DRY): a pack whose vol0 is NaN in every cell contributes exactly 0
// whatever theta/S hold (rho*NaN is skipped), so its lanes neither load nor compute; whole
// 64/128-byte lines of land or sub-bottom cells then never leave HBM.  Same bits out.
// FMA (MLX_FLAG_FMA): FusedOps arithmetic (eos_device.hpp; FusedTailOps on float32 input in
// numpy's mixed precision), not bit-identical to numpy.
// ------------------------------------------------------------------------------------
constexpr int kVarSteric = 0, kVarHalo = 1, kVarThermo = 2, kVarAll = 3;

// P3D (fast kernels only): the pressure is a (z,y,x) FIELD (MLX_P_FULL3D: `patm` given as a (yh,xh)
// DataArray, steric.py:58-60,96) -- each thread keeps its cells' pressures in registers beside vol0,
// read once with the same 16-byte loads; everything else is the z-profile kernel.
template <typename TIn, int VEC, int U, int VAR, int MODE, bool GENERIC, bool SKIP, bool FMA,
          bool P3D = false>
__global__ __launch_bounds__(kBlock) void k_steric_global(
    const TIn* __restrict__ T, const TIn* __restrict__ S, const TIn* __restrict__ T0,
    const TIn* __restrict__ S0, const double* __restrict__ vol0, const double* __restrict__ p,
    int p_mode, int eos, int nt, int t_chunk, int64_t plane, int64_t t_stride_T,
    int64_t t_stride_S, double* __restrict__ partials, int64_t nblk_total) {
  constexpr int NOUT = (VAR == kVarAll) ? 4 : 1;
  constexpr int NTC = (VAR == kVarAll) ? kNTC / 2 : kNTC;  // LDS: NOUT*NTC*kBlock doubles
  constexpr bool STREAM_T = (VAR != kVarHalo), STREAM_S = (VAR != kVarThermo);
  constexpr bool HELD_T = (VAR == kVarHalo || VAR == kVarAll);
  constexpr bool HELD_S = (VAR == kVarThermo || VAR == kVarAll);
  typedef typename PolyType<MODE>::type R;
  // arithmetic groups: pairs of adjacent cells on float2 in the faithful float32 fast path
  // (eos_device.hpp, PolyVec), single cells otherwise
  typedef typename PolyVec<MODE, GENERIC ? 1 : VEC>::type RV;
  constexpr int W = Lanes<RV>::n, G = VEC / W;
  // guarded policies in the fast kernels: the scale-free reciprocal with a per-wave fallback to
  // the IEEE division (eos_device.hpp, quotients<>); same bits as ExactOps in exact mode.  The
  // all-variants kernel sits at the 256-VGPR / 2-waves-per-SIMD edge and every form of the guard
  // measured (per batch, per step with reloaded operands) pushed it over: it keeps the IEEE
  // division in exact mode -- the same bits as the guarded single-variant launches.
  constexpr bool GUARD = !GENERIC && VAR != kVarAll;
  // the predicated skipna accumulate where the kernel is VALU-bound (eos_device.hpp add_skipna)
  constexpr bool PRED = !(VAR == kVarSteric && sizeof(TIn) == 8);
  typedef typename std::conditional<
      FMA, typename FusedFor<MODE>::type,
      typename std::conditional<GUARD, typename ExactFastFor<MODE>::type, ExactOps>::type>::type Ops;
  __shared__ double red[NOUT][NTC][kBlock];

  const int tid = threadIdx.x;
  const int z = blockIdx.y;
  const int nz = gridDim.y;
  // blockIdx.z = time chunk (slowest grid dimension): the resident blocks all work inside one
  // window of t_chunk time steps instead of drifting over the whole record (+3.7 % measured
  // at nt=120, scripts/tune_k1.hip), at the price of re-reading vol0 once per chunk.
  const int tb = blockIdx.z * t_chunk;
  const int te = (tb + t_chunk < nt) ? (tb + t_chunk) : nt;
  const int64_t bx = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t blk = (int64_t)blockIdx.y * gridDim.x + bx;  // partial slot = tile position
  const int64_t tile0 = bx * (kBlock * VEC * U);
  const int64_t zoff = (int64_t)z * plane;

  int64_t off[U];  // offset of pack u inside a (z,y,x) slab; clamped when past the plane
  double vol[U][VEC];
  double pc[U][VEC];
  bool alive[U];   // SKIP: false when every cell of the pack is dry (always true otherwise)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int64_t i = tile0 + ((int64_t)u * kBlock + tid) * VEC;
    const bool valid = (i + VEC <= plane);  // plane % VEC == 0 is a host-side precondition
    i = valid ? i : 0;
    off[u] = zoff + i;
    Pack<double, VEC> v = load_pack<double, VEC>(vol0 + off[u]);
    bool any_wet = false;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      vol[u][k] = valid ? v.v[k] : canonical_nan();
      // The fused accumulate (eos_device.hpp accumulate<>: c = fma(rho, vol, c) unless rho or vol
      // is NaN) never forms the product the reference tests (derived.py:435: the skipna sum of
      // rho * volcello): rho = +-inf on a cell of ZERO volume is NaN there, and skipped, where the
      // fma would poison the sum.  A zero-volume cell never changes the reference's sum -- its term
      // is +-0 or a skipped NaN -- so it is made a NaN-volume cell HERE, once per block and cell
      // (the volume is time-invariant: nothing in the time loop), and the fused sum is the
      // reference's for every operand class but rho = +-0 on a cell of INFINITE volume
      // (tests/test_gpu_kernels.py::test_fused_sum_skips_what_the_reference_skips).
      if constexpr (Ops::contracts && MLX_TUNE_FMA_ACC)
        vol[u][k] = (vol[u][k] == 0.0) ? canonical_nan() : vol[u][k];
      any_wet = any_wet || !is_nan(vol[u][k]);
    }
    alive[u] = !SKIP || any_wet;
    if ((GENERIC && p_mode == MLX_P_FULL3D) || P3D) {
      Pack<double, VEC> q = load_pack<double, VEC>(p + off[u]);
#pragma unroll
      for (int k = 0; k < VEC; ++k) pc[u][k] = q.v[k];
    }
  }
  static_assert(!(P3D && GENERIC), "P3D is the fast kernels' form of MLX_P_FULL3D");
  static_assert(!Ops::fused || W == 1, "pressure folding is per cell");
  double pz = 0.0;
  if ((!GENERIC && !P3D) || p_mode == MLX_P_ZPROF) pz = p[z];
  if (GENERIC && p_mode == MLX_P_SCALAR) pz = p[0];
  // pressure of cell k of pack u; FusedOps (float64 polynomial) folds it into B0
  auto p_at = [&](int u, int k) -> double { return P3D ? pc[u][k] : pz; };
  auto pfold_at = [&](int u, int g) -> RV {
    RV f = RV{};
    if constexpr (Ops::fused) f = (R)p_at(u, g * W);
    return f;
  };
  // does the pressure veto the fast quotient (a whole level's, or -- P3D -- any lane's of pack u)?
  lanemask_t pbad[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    pbad[u] = P3D ? 0 : Ops::p_unsafe(pz);
    if constexpr (P3D) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) pbad[u] |= Ops::p_unsafe_lanes(pc[u][k]);
    }
  }

  // held fields: read once; fast path keeps their PART of the polynomial, generic the values
  TPart<RV> t0p[(HELD_T && !GENERIC) ? U : 1][G];
  SPart<RV> s0p[(HELD_S && !GENERIC) ? U : 1][G];
  TIn t0v[(HELD_T && GENERIC) ? U : 1][VEC], s0v[(HELD_S && GENERIC) ? U : 1][VEC];
  if constexpr (HELD_T) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      Pack<TIn, VEC> h = {};
      if (alive[u]) h = load_field<MODE, true, TIn, VEC, true>(T0, off[u]);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if constexpr (GENERIC) t0v[u][k] = h.v[k];
      }
      if constexpr (!GENERIC) {
#pragma unroll
        for (int g = 0; g < G; ++g) t0p[u][g] = t_part_m<MODE, Ops, RV>(Lanes<RV>::make(&h.v[g * W]));
      }
    }
  }
  if constexpr (HELD_S) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      Pack<TIn, VEC> h = {};
      if (alive[u]) h = load_field<MODE, false, TIn, VEC, true>(S0, off[u]);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if constexpr (GENERIC) s0v[u][k] = h.v[k];
      }
      if constexpr (!GENERIC) {
#pragma unroll
        for (int g = 0; g < G; ++g)
          s0p[u][g] = s_part_m<MODE, Ops, RV>(Lanes<RV>::make(&h.v[g * W]), pfold_at(u, g));
      }
    }
  }

  // Register double buffering of the streams -- except in the float64 all-variants kernel, whose
  // 9 held doubles per cell leave no room for a second set of packs: it would drop to one wave
  // per SIMD.  That kernel prefetches THROUGH LDS instead (GLDS): the next step's packs are fetched
  // by global_load_lds_dwordx4 -- an asynchronous copy HBM -> LDS with no VGPR destination
  // (cdna_hip_programming.md section 5: "in a kernel already at the VGPR cap ... glds") -- while
  // the current step, read out of the same LDS buffer into the one register set, is evaluated.
  // Each wave reads back exactly the 1 KiB slices its own lanes fetched (destination = wave-uniform
  // base + lane*16), so no barrier is involved: vmcnt(0) before the read-out, lgkmcnt(0) before the
  // buffer is overwritten by the next step's copies.
  constexpr bool PREFETCH = !(VAR == kVarAll && sizeof(TIn) == 8 && !GENERIC);
  // (theta and salinity of different dtypes: the float32 field's pack is 8 bytes, not the 16 of an
  //  LDS-DMA slot -- that rare all-variants kernel simply loads each step's packs when it needs them)
  constexpr bool GLDS = !PREFETCH && !IsMixed<MODE>::value;
  constexpr bool DIRECT = !PREFETCH && !GLDS;
  static_assert(!GLDS || sizeof(TIn) * VEC == 16, "one 16-byte LDS-DMA per pack");
  __shared__ f4_t stage[GLDS ? 2 * U : 1][kBlock];  // [field*U + u][thread]: 32 KiB
  Pack<TIn, VEC> curT[U], curS[U], nxtT[PREFETCH ? U : 1], nxtS[PREFETCH ? U : 1];
  if constexpr (GLDS) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (alive[u]) {
        lds_dma16(T + (int64_t)tb * t_stride_T + off[u], &stage[u][tid & ~63]);
        lds_dma16(S + (int64_t)tb * t_stride_S + off[u], &stage[U + u][tid & ~63]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < (PREFETCH ? U : 0); ++u) {
    nxtT[u] = {};
    nxtS[u] = {};
    if (alive[u]) {
      if (STREAM_T) nxtT[u] = load_field<MODE, true, TIn, VEC, true>(T, (int64_t)tb * t_stride_T + off[u]);
      if (STREAM_S) nxtS[u] = load_field<MODE, false, TIn, VEC, true>(S, (int64_t)tb * t_stride_S + off[u]);
    }
  }

  for (int t = tb; t < te; ++t) {
    if constexpr (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's packs have landed in LDS
#pragma unroll
      for (int u = 0; u < U; ++u) {  // (a dry pack's slot holds stale data: its lanes skip the sums)
        __builtin_memcpy(&curT[u], &stage[u][tid], 16);
        __builtin_memcpy(&curS[u], &stage[U + u][tid], 16);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read out before the buffer is reused
      if (t + 1 < te) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (alive[u]) {
            lds_dma16(T + (int64_t)(t + 1) * t_stride_T + off[u], &stage[u][tid & ~63]);
            lds_dma16(S + (int64_t)(t + 1) * t_stride_S + off[u], &stage[U + u][tid & ~63]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < (PREFETCH ? U : 0); ++u) {
      if (STREAM_T) curT[u] = nxtT[u];
      if (STREAM_S) curS[u] = nxtS[u];
    }
    if constexpr (DIRECT) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        curT[u] = {};
        curS[u] = {};
        if (alive[u]) {
          curT[u] = load_field<MODE, true, TIn, VEC, true>(T, (int64_t)t * t_stride_T + off[u]);
          curS[u] = load_field<MODE, false, TIn, VEC, true>(S, (int64_t)t * t_stride_S + off[u]);
        }
      }
    }
    if (PREFETCH && t + 1 < te) {  // issue the next step's loads before this step's arithmetic
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (alive[u]) {
          if (STREAM_T)
            nxtT[u] = load_field<MODE, true, TIn, VEC, true>(T, (int64_t)(t + 1) * t_stride_T + off[u]);
          if (STREAM_S)
            nxtS[u] = load_field<MODE, false, TIn, VEC, true>(S, (int64_t)(t + 1) * t_stride_S + off[u]);
        }
      }
    }
    double c[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) c[o] = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (SKIP && !alive[u]) continue;  // adds exactly nothing: c + 0.0 == c
      constexpr int NR = NOUT > 1 ? 3 : 1;
      double rho[NR][VEC];
      if constexpr (GENERIC) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          double pp = pz;
          if (p_mode == MLX_P_FULL3D) pp = pc[u][k];
          if (p_mode == MLX_P_FULL4D) pp = p[((int64_t)t * nz) * plane + off[u] + k];
          const TIn tv = STREAM_T ? curT[u].v[k] : t0v[u][k];
          const TIn sv = STREAM_S ? curS[u].v[k] : s0v[u][k];
          rho[0][k] = eos_eval<MODE, TIn, Ops>(eos, kDensity, tv, sv, pp);
          if constexpr (VAR == kVarAll) {
            rho[1][k] = eos_eval<MODE, TIn, Ops>(eos, kDensity, tv, s0v[u][k], pp);
            rho[2][k] = eos_eval<MODE, TIn, Ops>(eos, kDensity, t0v[u][k], sv, pp);
          }
        }
      } else if constexpr (VAR == kVarAll) {
        // three quotients per cell (unguarded policies in this kernel, see Ops above); summed group
        // by group -- the same cell order as below, fewer densities live at a time
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const TPart<RV> a = t_part_m<MODE, Ops, RV>(Lanes<RV>::make(&curT[u].v[g * W]));
          const SPart<RV> b = s_part_m<MODE, Ops, RV>(Lanes<RV>::make(&curS[u].v[g * W]), pfold_at(u, g));
          double r3[3][W];
          static_assert(!(P3D && VAR == kVarAll), "the all-variants kernel has no register to spare");
          wright_combine_lanes<Ops, RV>(a, b, pz, r3[0]);
          wright_combine_lanes<Ops, RV>(a, s0p[u][g], pz, r3[1]);
          wright_combine_lanes<Ops, RV>(t0p[u][g], b, pz, r3[2]);
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const int k = g * W + w;
#pragma unroll
            for (int o = 0; o < 3; ++o) accumulate<Ops, PRED>(c[o], r3[o][w], vol[u][k]);
            // extension: heat content -- product, then sum, whatever the density's policy (the row
            // must not depend on how rho is evaluated)
            accumulate<ExactOps, PRED>(c[3], (double)curT[u].v[k], vol[u][k]);
          }
        }
        continue;
      } else {
        // one guarded batch of quotients per pack (VEC cells)
        double num[VEC], den[VEC];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          TPart<RV> a;
          SPart<RV> b;
          if constexpr (STREAM_T) a = t_part_m<MODE, Ops, RV>(Lanes<RV>::make(&curT[u].v[g * W]));
          if constexpr (STREAM_S)
            b = s_part_m<MODE, Ops, RV>(Lanes<RV>::make(&curS[u].v[g * W]), pfold_at(u, g));
          double pg[W];
#pragma unroll
          for (int w = 0; w < W; ++w) pg[w] = p_at(u, g * W + w);
          if constexpr (VAR == kVarSteric)
            wright_numden_lanes<Ops, RV>(a, b, pg, &num[g * W], &den[g * W]);
          if constexpr (VAR == kVarHalo)
            wright_numden_lanes<Ops, RV>(t0p[u][g], b, pg, &num[g * W], &den[g * W]);
          if constexpr (VAR == kVarThermo)
            wright_numden_lanes<Ops, RV>(a, s0p[u][g], pg, &num[g * W], &den[g * W]);
        }
        quotients<Ops, VEC>(num, den, rho[0], pbad[u]);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {  // cells in ascending order: the order of summation is fixed
#pragma unroll
        for (int o = 0; o < NR; ++o) accumulate<Ops, PRED>(c[o], rho[o][k], vol[u][k]);  // derived.py:435
        if constexpr (VAR == kVarAll)  // extension: heat-content integrand theta*vol0 (see above)
          accumulate<ExactOps, PRED>(c[3], (double)curT[u].v[k], vol[u][k]);
      }
    }
    const int row = (t - tb) % NTC;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) red[o][row][tid] = c[o];
    if (row == NTC - 1 || t == te - 1) {
      __syncthreads();
      const int wave = tid >> 6, lane = tid & 63;
      for (int q = wave; q < NOUT * (row + 1); q += kBlock / 64) {
        const int o = q / (row + 1), r = q % (row + 1);
        double v = ((red[o][r][lane] + red[o][r][lane + 64]) + red[o][r][lane + 128]) +
                   red[o][r][lane + 192];
        v = wave_sum(v);
        if (lane == 0) partials[((int64_t)o * nt + (t - row + r)) * nblk_total + blk] = v;
      }
      __syncthreads();
    }
  }
}

// partials[row][0..n) -> out[row]; one block per row, fixed order
__global__ __launch_bounds__(kBlock) void k_reduce_rows(const double* __restrict__ partials,
                                                        int64_t n, double* __restrict__ out) {
  __shared__ double red[kBlock];
  const double* row = partials + (int64_t)blockIdx.x * n;
  double c = 0.0;
  // eight loads in flight, then the eight adds IN THE ORDER of the plain loop (the same bits): one
  // load per dependent add left a 24-row launch -- a time chunk of the tiled walk -- at 85 us for
  // 11 MB (latency, not bandwidth), paid once per chunk (profiles/r06_forced_collective_trace.txt)
  int64_t i = threadIdx.x;
  for (; i + 7 * (int64_t)kBlock < n; i += 8 * (int64_t)kBlock) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = row[i + (int64_t)k * kBlock];
#pragma unroll
    for (int k = 0; k < 8; ++k) c += v[k];
  }
  for (; i < n; i += kBlock) c += row[i];
  red[threadIdx.x] = c;
  __syncthreads();
#pragma unroll
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------------------
// standalone calc_masso on a materialised rho (derived.py:435-438): per-block partial
// of sum(rho*vol) [skipna] for one time step; grid = (nblk, nt)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_masso_partial(const double* __restrict__ rho,
                                                          const double* __restrict__ vol,
                                                          int64_t n3, int64_t vol_t_stride,
                                                          double* __restrict__ partials) {
  __shared__ double red[kBlock / 64];
  const int64_t t = blockIdx.y;
  const double* r = rho + t * n3;
  const double* v = vol + t * vol_t_stride;
  double c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n3;
       i += (int64_t)gridDim.x * kBlock) {
    const double term = r[i] * v[i];
    c += is_nan(term) ? 0.0 : term;
  }
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0)
    partials[t * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// skipna sum, stage 1: grid-stride, per-block partial.  VEC2: 16-byte nt loads (x 16-byte aligned,
// n even) -- the same streaming access as K1, so bench.py also uses it as the box's read-ceiling
// probe; otherwise scalar loads.
template <bool VEC2>
__global__ __launch_bounds__(kBlock) void k_nansum_partial(const double* __restrict__ x, int64_t n,
                                                           double* __restrict__ partials) {
  __shared__ double red[kBlock / 64];
  double c = 0.0;
  if constexpr (VEC2) {
    const int64_t n2 = n / 2;
    double c1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += (int64_t)gridDim.x * kBlock) {
      const Pack<double, 2> v = load_pack<double, 2, true>(x + 2 * i);
      c += is_nan(v.v[0]) ? 0.0 : v.v[0];
      c1 += is_nan(v.v[1]) ? 0.0 : v.v[1];
    }
    c += c1;
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kBlock) {
      const double v = x[i];
      c += is_nan(v) ? 0.0 : v;
    }
  }
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// Streaming probes: the box's ceiling for each read:write mix of the fused kernels, with no
// arithmetic to speak of.  NIN streams of TIn in (16-byte nt loads: 2 doubles or 4 floats per lane)
// and, if WRITE, one float64 stream out (16-byte stores): K1 is 2 in / read-only, a held-field
// global pass 1 in / read-only, the local pass with delta_rho 2 in / 1 out (16 B + 8 B per cell at
// float64, 8 B + 8 B at float32), a held-field local pass 1 in / 1 out, the eta-only passes
// read-only.  Without WRITE the values are summed into a per-thread sink that is stored only if it
// equals a value no data produces (keeps the loads alive, writes nothing).
//
// Shape (round 5, scripts/tune_probe.hip, profiles/r05_tune_probe*.log): ONE TILE PER BLOCK -- a
// block loads U packs per thread and stream, stores, and ends -- with U and the store policy picked
// per mix from the sweep.  Round 4's shape (a grid-stride loop, 65536 blocks, one pack in flight
// per thread) read 3.7 TB/s for "1 x float32 in, 1 x float64 out" on a box where this one reads
// 6.0, and the kernel it was meant to bound ran at 5.1: a probe slower than the kernel is not a
// ceiling.  (Beyond 2^23 tiles the blocks stride over the tiles: a HIP grid holds < 2^32 threads.)
constexpr int64_t kProbeMaxBlocks = (int64_t)1 << 23;
constexpr int64_t kK2MaxBlocks1D = ((int64_t)1 << 24) - 1;  // x kBlock (256) threads < 2^32
// VEC: elements per lane and pack -- 16 bytes' worth, except float32 in / float64 out, where TWO
// floats per lane (an 8-byte load, ONE 16-byte store: the access widths of K2's two-column float32
// shape) read 6.6 / 6.1 TB/s where four floats and two stores per lane read 5.9 / 5.7
// (profiles/r05_tune_probe_f32_out.log) -- and the kernels they bound had reached 5.95 / 5.61.
template <typename TIn, int NIN, bool WRITE, int U, bool NTS, int VEC = 16 / sizeof(TIn)>
__global__ __launch_bounds__(kBlock) void k_stream_probe_mix(const TIn* __restrict__ a,
                                                             const TIn* __restrict__ b,
                                                             int64_t npacks,
                                                             double* __restrict__ out) {
  double sink = 0.0;
  const int64_t ntiles = (npacks + kBlock * U - 1) / (kBlock * U);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * (kBlock * U) + threadIdx.x;
    Pack<TIn, VEC> x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + (int64_t)u * kBlock;
      if (i < npacks) {
        x[u] = load_pack<TIn, VEC, true>(a + VEC * i);
        if constexpr (NIN == 2) y[u] = load_pack<TIn, VEC, true>(b + VEC * i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + (int64_t)u * kBlock;
      if (i < npacks) {
        Pack<double, VEC> r;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          r.v[k] = (double)x[u].v[k];
          if constexpr (NIN == 2) r.v[k] += (double)y[u].v[k];
        }
        if constexpr (WRITE) {
          store_pack<VEC, NTS>(out + VEC * i, r);
        } else {
#pragma unroll
          for (int k = 0; k < VEC; ++k) sink += r.v[k];
        }
      }
    }
  }
  if constexpr (!WRITE) {
    if (sink == 0x1.23456789abcdep+1000) out[0] = sink;
  }
}

// VALU issue-rate probe: 8 independent chains of v_fma_f64 per thread and nothing else -- the
// box's float64 vector-ALU ceiling in lane-instructions per second, against which bench.py prices
// the kernels that are bound by instruction issue rather than by HBM (every float32 kernel, the
// exact held-field sums, the one-pass kernels).  kValuProbeBlocks x 256 threads x 8 x iters fmas.
constexpr int kValuProbeBlocks = 8192;
__global__ __launch_bounds__(kBlock) void k_valu_probe(double* __restrict__ out, int iters,
                                                       double x, double y) {
  double a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (double)(threadIdx.x + k);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], x, y);
  }
  double sink = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) sink += a[k];
  if (sink == 0x1.23456789abcdep+1000) out[0] = sink;  // never: keeps the chains alive
}

// ------------------------------------------------------------------------------------
// K0: pointwise EOS map.  grid = (ceil(plane/(kBlock*VEC*U)), nz, nt)
// ------------------------------------------------------------------------------------
template <typename TIn, int VEC, int U, int MODE, int FUNC, bool GENERIC, bool FMA = false,
          bool P3D = false>
__global__ __launch_bounds__(kBlock) void k_eos_map(const TIn* __restrict__ T,
                                                    const TIn* __restrict__ S,
                                                    const double* __restrict__ p, int p_mode,
                                                    int eos, int func, int64_t nz, int64_t plane,
                                                    int64_t t_stride_T, int64_t t_stride_S,
                                                    int64_t t_base, double aux,
                                                    double* __restrict__ out) {
  // the fast density kernels use the guarded policies (scale-free reciprocal, eos_device.hpp)
  typedef typename std::conditional<
      FMA, typename FusedFor<MODE>::type,
      typename std::conditional<!GENERIC && FUNC == kDensity, typename ExactFastFor<MODE>::type,
                                ExactOps>::type>::type Ops;
  const int z = blockIdx.y;
  const int64_t t = t_base + blockIdx.z;
  const int64_t tile0 = (int64_t)blockIdx.x * (kBlock * VEC * U);
  const int64_t zoff = (int64_t)z * plane;
  double pz = 0.0;
  if ((!GENERIC && !P3D) || p_mode == MLX_P_ZPROF) pz = p[z];
  if (GENERIC && p_mode == MLX_P_SCALAR) pz = p[0];
  Pack<TIn, VEC> a[U], b[U];
  Pack<double, VEC> pq[U];  // P3D: the cells' own pressures (a (z,y,x) field), 16-byte loads
  int64_t off[U];
  bool valid[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = tile0 + ((int64_t)u * kBlock + threadIdx.x) * VEC;
    valid[u] = (i + VEC <= plane);
    off[u] = zoff + (valid[u] ? i : 0);
    a[u] = load_pack<TIn, VEC, true>(T + t * t_stride_T + off[u]);
    b[u] = load_pack<TIn, VEC, true>(S + t * t_stride_S + off[u]);
    if constexpr (P3D) pq[u] = load_pack<double, VEC>(p + off[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    Pack<double, VEC> r;
    if constexpr (GENERIC) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        double pp = pz;
        if (p_mode == MLX_P_FULL3D) pp = p[off[u] + k];
        if (p_mode == MLX_P_FULL4D) pp = p[t * nz * plane + off[u] + k];
        r.v[k] = eos_eval<MODE, TIn, Ops>(eos, func, a[u].v[k], b[u].v[k], pp, aux);
      }
    } else if constexpr (FUNC == kDensity) {
      // float32 faithful: the polynomial on float2 pairs (see K1), otherwise cell by cell; one
      // guarded batch of quotients per pack
      typedef typename PolyType<MODE>::type R;
      typedef typename PolyVec<MODE, VEC>::type RV;
      constexpr int W = Lanes<RV>::n;
      static_assert(!Ops::fused || W == 1, "pressure folding is per cell");
      double num[VEC], den[VEC];
      lanemask_t pbad = P3D ? 0 : Ops::p_unsafe(pz);
#pragma unroll
      for (int k = 0; k < VEC; k += W) {
        double pg[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          pg[w] = P3D ? pq[u].v[k + w] : pz;
          if constexpr (P3D) pbad |= Ops::p_unsafe_lanes(pg[w]);
        }
        RV pfold = RV{};
        if constexpr (Ops::fused) pfold = (R)pg[0];
        wright_numden_lanes<Ops, RV>(t_part_m<MODE, Ops, RV>(Lanes<RV>::make(&a[u].v[k])),
                                     s_part_m<MODE, Ops, RV>(Lanes<RV>::make(&b[u].v[k]), pfold), pg,
                                     &num[k], &den[k]);
      }
      quotients<Ops, VEC>(num, den, r.v, pbad);
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        r.v[k] = eos_eval<MODE, TIn, Ops>(kWright, FUNC, a[u].v[k], b[u].v[k],
                                          P3D ? pq[u].v[k] : pz);
    }
    if (valid[u]) store_pack<VEC, true>(out + t * nz * plane + off[u], r);
  }
}

// rho0m = where(vol0 notnull, rho0, NaN)
__global__ __launch_bounds__(kBlock) void k_fold_mask(const double* __restrict__ rho0,
                                                      const double* __restrict__ vol0, int64_t n,
                                                      double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock)
    out[i] = is_nan(vol0[i]) ? canonical_nan() : rho0[i];
}

// calc_dz's default-argument core, derived.py:295-318 with top=0, bottom=None
__device__ __forceinline__ double dz_default(double depth, double ztop, double zbot) {
  const double d = is_nan(depth) ? 0.0 : depth;  // fillna(0.0)
  const double dz_field = zbot - ztop;
  double part = d - ztop;
  part = (part < 0.0) ? 0.0 : part;
  double result = (part < dz_field) ? part : dz_field;  // np.minimum (no NaN possible here)
  part = zbot - 0.0;
  part = (part < 0.0) ? 0.0 : part;
  result = (part < result) ? part : result;
  return result;
}

// ------------------------------------------------------------------------------------
// K2: fused EOS + delta_rho + dz-weighted column integral.
//
// grid = (ceil(plane/(kBlock*VEC)), ceil(nt/NTI)).  A thread owns VEC adjacent columns
// and NTI consecutive time steps: z is the outer (sequential, as numpy's axis reduce)
// loop, the NTI time steps are unrolled inside it with their column sums in registers,
// so rho0m / dz are read once per z and reused NTI times.  NTI*VEC = 32 column sums per
// thread (64 VGPRs) + 2*NTI loads in flight: ~240 VGPRs, 2 waves/SIMD -- measured faster
// (scripts/tune_k2.hip) than NTI=8 at 4 waves/SIMD: half the rho0m re-reads and twice the
// bytes in flight per wave.  theta/S loads and the delta_rho stores use the nt policy.
// ------------------------------------------------------------------------------------
// VAR (steric.py:115-125), as in K1: 0 steric; 1 halosteric (theta held: T is the (z,y,x) slab,
// stride 0); 2 thermosteric (S held); 3 = ALL THREE in one pass: theta/S are read once, the held
// slabs T0/S0 once per level, and three delta_rho / eta fields are written, variant v at
// out + v*variant_stride (v = 0 steric, 1 thermosteric, 2 halosteric, the order of K1's rows):
// 16 B read + 3*8 B written per cell instead of 3 x (16 or 8 read + 8 written).  Each field is
// bit-identical to its single-variant launch (same arithmetic tree, z ascending).
// P3D (fast kernels): the pressure is a (z,y,x) field, read per level like rho0m (see K1).
// LDSACC (round 5 experiment, -DMLX_TUNE_K2_LDSACC=1; off in the product): the column sums of the
// NTI time steps live in LDS ([step][column][thread]: a wave's 64 lanes hit 64 consecutive doubles,
// conflict-free; every thread owns its slots, no barrier) instead of VGPRs, so that one thread can
// take 32 steps of one column -- north_star's "LDS-staged z-column tiles", VERDICT r4 next #5.
template <typename TIn, int VEC, int NTI, int VAR, int MODE, bool GENERIC, bool SKIP, bool FMA,
          bool P3D = false, bool LDSACC = false>
__global__ __launch_bounds__(kBlock) void k_steric_local(
    const TIn* __restrict__ T, const TIn* __restrict__ S, const TIn* __restrict__ T0,
    const TIn* __restrict__ S0, const double* __restrict__ rho0m,
    const double* __restrict__ vol0_surface, const double* __restrict__ dz,
    const double* __restrict__ z_i, const double* __restrict__ deptho,
    const double* __restrict__ p, int p_mode, int eos, double neg_inv_rhozero, int nt, int nz,
    int64_t plane, int64_t t_stride_T, int64_t t_stride_S, double* __restrict__ drho_out,
    int64_t drho_vstride, double* __restrict__ eta_out, int64_t eta_vstride, int ntb_major) {
  constexpr int NOUT = (VAR == kVarAll) ? 3 : 1;
  constexpr bool STREAM_T = (VAR != kVarHalo), STREAM_S = (VAR != kVarThermo);
  constexpr bool HELD_T = (VAR == kVarHalo || VAR == kVarAll);
  constexpr bool HELD_S = (VAR == kVarThermo || VAR == kVarAll);
  typedef typename PolyType<MODE>::type R;
  typedef typename PolyVec<MODE, GENERIC ? 1 : VEC>::type RV;  // float2 pairs: see K1
  constexpr int W = Lanes<RV>::n, G = VEC / W;
  typedef typename std::conditional<
      FMA, typename FusedFor<MODE>::type,
      typename std::conditional<GENERIC, ExactOps, typename ExactFastFor<MODE>::type>::type>::type Ops;
  // add_skipna form: +2-3 % at float32; at float64 neutral for the steric pass (round 3) and 0-2 %
  // for the held-field passes (round 4, profiles/r04_tune_k2_pred_held64.log: inside the scatter)
  constexpr bool PRED = sizeof(TIn) == 4;
  // Block -> (column tile, time block).  ntb_major > 0 (round 5): a 1-D grid in which the ntb_major
  // TIME BLOCKS OF ONE COLUMN TILE are consecutive workgroups of one XCD (workgroups are dealt
  // round-robin over the 8 XCDs: ids b, b+8, b+16, ... share one, and with it an L2), so that the
  // siblings run side by side and the time-invariant operands every one of them reads -- rho0m, dz,
  // the held field: 8-16 B per cell and time block -- come from HBM once and from L2 for the other
  // ntb_major-1.  With the time blocks on grid.y (ntb_major == 0) a tile's next time block starts
  // a whole plane of blocks later and re-reads those lines from HBM.  Speed and traffic only:
  // which thread sums which column in which order does not change.
  int64_t tile;
  int tb;
  if (ntb_major > 0) {
    const int64_t q = blockIdx.x >> 3;
    tb = (int)(q % ntb_major);
    tile = (q / ntb_major) * 8 + (blockIdx.x & 7);
  } else {
    tile = xcd_remap(blockIdx.x, gridDim.x);
    tb = blockIdx.y;
  }
  const int64_t col = (tile * kBlock + threadIdx.x) * VEC;
  if (col + VEC > plane) return;  // whole packs only; no barrier below
  const int t0 = tb * NTI;
  const int64_t n3 = (int64_t)nz * plane;

  constexpr int NREG = LDSACC ? 1 : NTI;  // (LDSACC: the register array shrinks to a placeholder)
  double acc[NOUT][NREG][VEC];
  __shared__ double sacc[LDSACC ? NTI * VEC * kBlock : 1];
  static_assert(!LDSACC || NOUT == 1, "LDS column sums: single-variant passes only");
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int j = 0; j < NREG; ++j)
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[o][j][k] = 0.0;
  if constexpr (LDSACC) {
#pragma unroll
    for (int i = 0; i < NTI * VEC; ++i) sacc[i * kBlock + threadIdx.x] = 0.0;
  }

  Pack<double, VEC> depth;
  if (dz == nullptr) depth = load_pack<double, VEC>(deptho + col);
  double pz = 0.0;
  if (GENERIC && p_mode == MLX_P_SCALAR) pz = p[0];

  // rho0m of level z+1 is fetched while level z is processed, so that the dry test (SKIP) and the
  // theta/S loads of a level never wait for a dependent load
  Pack<double, VEC> r0n = load_pack<double, VEC>(rho0m + col);

  for (int z = 0; z < nz; ++z) {
    const int64_t off = (int64_t)z * plane + col;
    // The time strides pass through an empty asm in every iteration: the compiler then computes each
    // load's and store's scalar offset (t0+j)*stride where it is used instead of hoisting
    // 3*NTI loop-invariant 64-bit offsets out of the z loop -- which needs more scalar registers
    // than the wave has (the float64 kernel spilled ~150 of them into VGPR lanes, v_writelane /
    // v_readlane + hazard nops in the hot loop: scripts/isa_count.py).
    int64_t sT = t_stride_T, sS = t_stride_S, sD = n3;
    asm volatile("" : "+s"(sT), "+s"(sS), "+s"(sD));
    int64_t oT = (int64_t)t0 * sT, oS = (int64_t)t0 * sS, oD = (int64_t)t0 * sD;  // of step t0 + j
    int nvalid = nt - t0;  // steps of this (possibly ragged) chunk: likewise compared where used
    asm volatile("" : "+s"(nvalid));
    const Pack<double, VEC> r0 = r0n;
    if (z + 1 < nz) r0n = load_pack<double, VEC>(rho0m + off + plane);
    Pack<double, VEC> dzv;
    if (dz != nullptr) {
      dzv = load_pack<double, VEC>(dz + off);
    } else {
      const double ztop = z_i[z], zbot = z_i[z + 1];
#pragma unroll
      for (int k = 0; k < VEC; ++k) dzv.v[k] = dz_default(depth.v[k], ztop, zbot);
    }
    if ((!GENERIC && !P3D) || p_mode == MLX_P_ZPROF) pz = p[z];
    Pack<double, VEC> pfull;
    if ((GENERIC && p_mode == MLX_P_FULL3D) || P3D) pfull = load_pack<double, VEC>(p + off);
    static_assert(!(P3D && GENERIC), "P3D is the fast kernels' form of MLX_P_FULL3D");
    static_assert(!Ops::fused || W == 1, "pressure folding is per cell");
    // the pressure's veto of the fast quotient: the level's, or -- P3D -- any lane's of this pack
    lanemask_t pbad = P3D ? 0 : Ops::p_unsafe(pz);
    if constexpr (P3D) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) pbad |= Ops::p_unsafe_lanes(pfull.v[k]);
    }

    // SKIP (MLX_FLAG_SKIP_DRY): where rho0m is NaN in every cell of the pack, delta_rho is NaN and
    // the column sum unchanged whatever theta/S hold -> their loads are masked off, same bits
    bool alive = true;
    if constexpr (SKIP) {
      alive = false;
#pragma unroll
      for (int k = 0; k < VEC; ++k) alive = alive || !is_nan(r0.v[k]);
    }

    // a held field is read once per level and -- fast path -- its part of the polynomial
    // evaluated once for the NTI time steps (eos_device.hpp TPart/SPart)
    Pack<TIn, VEC> hT = {}, hS = {};
    if (alive) {
      if (HELD_T) hT = load_field<MODE, true, TIn, VEC>(VAR == kVarAll ? T0 : T, off);
      if (HELD_S) hS = load_field<MODE, false, TIn, VEC>(VAR == kVarAll ? S0 : S, off);
    }
    RV pfold[G];  // FusedOps folds the pressure (the level's, or the cell's own) into B0
    double pg[G][W];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int w = 0; w < W; ++w) pg[g][w] = P3D ? pfull.v[g * W + w] : pz;
      pfold[g] = RV{};
      if constexpr (Ops::fused) pfold[g] = (R)pg[g][0];
    }
    TPart<RV> hTp[G];
    SPart<RV> hSp[G];
    if constexpr (!GENERIC && HELD_T) {
#pragma unroll
      for (int g = 0; g < G; ++g) hTp[g] = t_part_m<MODE, Ops, RV>(Lanes<RV>::make(&hT.v[g * W]));
    }
    if constexpr (!GENERIC && HELD_S) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        hSp[g] = s_part_m<MODE, Ops, RV>(Lanes<RV>::make(&hS.v[g * W]), pfold[g]);
    }

    Pack<TIn, VEC> a[NTI], b[NTI];
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
      a[j] = {};
      b[j] = {};
      if (alive && j < nvalid) {  // the ragged last chunk issues no surplus loads
        if (STREAM_T) a[j] = load_field<MODE, true, TIn, VEC, true>(T, oT + off);
        if (STREAM_S) b[j] = load_field<MODE, false, TIn, VEC, true>(S, oS + off);
      }
      oT += sT;
      oS += sS;
    }
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
      if (j < nvalid) {
        Pack<double, VEC> d[NOUT];
        // dry lanes (SKIP) run the same arithmetic on zeros: rho - NaN is NaN and the NaN term is
        // skipped, exactly as if theta/S had been loaded.  Only their LOADS are masked -- a
        // divergent store path would split every partly-dry line into two transactions.
        double rho[NOUT][VEC];
        if constexpr (GENERIC) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const TIn tv = STREAM_T ? a[j].v[k] : hT.v[k];
            const TIn sv = STREAM_S ? b[j].v[k] : hS.v[k];
            double pp = (p_mode == MLX_P_FULL3D) ? pfull.v[k] : pz;
            if (p_mode == MLX_P_FULL4D) pp = p[(int64_t)(t0 + j) * n3 + off + k];
            rho[0][k] = eos_eval<MODE, TIn, Ops>(eos, kDensity, tv, sv, pp);
            if constexpr (VAR == kVarAll) {
              rho[1][k] = eos_eval<MODE, TIn, Ops>(eos, kDensity, tv, hS.v[k], pp);
              rho[2][k] = eos_eval<MODE, TIn, Ops>(eos, kDensity, hT.v[k], sv, pp);
            }
          }
        } else {
          // one guarded batch of quotients (eos_device.hpp) per pack and time step: VEC cells x
          // NOUT variants
          double num[NOUT][VEC], den[NOUT][VEC];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            TPart<RV> tp;
            SPart<RV> sp;
            if constexpr (STREAM_T) tp = t_part_m<MODE, Ops, RV>(Lanes<RV>::make(&a[j].v[g * W]));
            if constexpr (STREAM_S)
              sp = s_part_m<MODE, Ops, RV>(Lanes<RV>::make(&b[j].v[g * W]), pfold[g]);
            double* const n0 = &num[0][g * W];
            double* const d0 = &den[0][g * W];
            const double* const pl = pg[g];
            if constexpr (VAR == kVarSteric) wright_numden_lanes<Ops, RV>(tp, sp, pl, n0, d0);
            if constexpr (VAR == kVarHalo) wright_numden_lanes<Ops, RV>(hTp[g], sp, pl, n0, d0);
            if constexpr (VAR == kVarThermo) wright_numden_lanes<Ops, RV>(tp, hSp[g], pl, n0, d0);
            if constexpr (VAR == kVarAll) {
              wright_numden_lanes<Ops, RV>(tp, sp, pl, n0, d0);
              wright_numden_lanes<Ops, RV>(tp, hSp[g], pl, &num[1][g * W], &den[1][g * W]);
              wright_numden_lanes<Ops, RV>(hTp[g], sp, pl, &num[2][g * W], &den[2][g * W]);
            }
          }
          quotients<Ops, NOUT * VEC>(&num[0][0], &den[0][0], &rho[0][0], pbad);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
            const double dr = rho[o][k] - r0.v[k];  // steric.py:152 (NaN where vol0 is NaN)
            d[o].v[k] = dr;
            const double term = dzv.v[k] * dr;          // steric.py:163
            if constexpr (LDSACC) {
              double c = sacc[(j * VEC + k) * kBlock + threadIdx.x];
              add_skipna<PRED>(c, term);
              sacc[(j * VEC + k) * kBlock + threadIdx.x] = c;
            } else {
              add_skipna<PRED>(acc[o][j][k], term);           // skipna, z ascending like numpy
            }
          }
        }
        if (drho_out != nullptr) {  // wave-uniform: the eta-only mode skips payload fix and store
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
              d[o].v[k] = is_nan(d[o].v[k]) ? canonical_nan() : d[o].v[k];  // canonical payload
            store_pack<VEC, true>(drho_out + o * drho_vstride + oD + off, d[o]);
          }
        }
      }
      oD += sD;
    }
  }
  const Pack<double, VEC> surf = load_pack<double, VEC>(vol0_surface + col);
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
      if (t0 + j < nt) {
        Pack<double, VEC> e;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const double sum = LDSACC ? sacc[(j * VEC + k) * kBlock + threadIdx.x] : acc[o][LDSACC ? 0 : j][k];
          e.v[k] = is_nan(surf.v[k]) ? canonical_nan() : neg_inv_rhozero * sum;
        }
        store_pack<VEC>(eta_out + o * eta_vstride + (int64_t)(t0 + j) * plane + col, e);
      }
    }
  }
}

// util.annual_average core (util.py:85-92): weighted mean over groups of L consecutive steps.
// grid = (ceil(n/(kBlock*VEC)), ngroups); streams x once, 16-byte nt loads when VEC == 2.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_group_weighted_mean(const double* __restrict__ x,
                                                                const double* __restrict__ w,
                                                                int64_t L, int64_t n,
                                                                double* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC;
  if (i + VEC > n) return;
  const int64_t g = blockIdx.y;
  double num[VEC], den[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) num[k] = den[k] = 0.0;
  for (int64_t j = 0; j < L; ++j) {
    const double wj = w[g * L + j];
    const Pack<double, VEC> v = load_pack<double, VEC, true>(x + (g * L + j) * n + i);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const bool bad = is_nan(v.v[k]);
      num[k] += (bad ? 0.0 : v.v[k]) * wj;
      den[k] += (bad ? 0.0 : 1.0) * wj;
    }
  }
  Pack<double, VEC> r;
#pragma unroll
  for (int k = 0; k < VEC; ++k) r.v[k] = num[k] / ((den[k] != 0.0) ? den[k] : canonical_nan());
  store_pack<VEC, true>(out + g * n + i, r);
}

// derived.calc_dz, derived.py:295-323 with explicit top/bottom/fraction
__global__ __launch_bounds__(kBlock) void k_calc_dz(const double* __restrict__ z_i,
                                                    const double* __restrict__ depth, int64_t nz,
                                                    int64_t plane, double top, double bottom,
                                                    int has_bottom, int fraction,
                                                    double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= plane) return;
  double d = depth[i];
  d = is_nan(d) ? 0.0 : d;
  if (has_bottom) d = (is_nan(bottom) || bottom < d) ? bottom : d;  // np.minimum(depth, bottom)
  for (int64_t z = 0; z < nz; ++z) {
    const double ztop = z_i[z], zbot = z_i[z + 1];
    const double dz_field = zbot - ztop;
    double part = d - ztop;
    part = (part < 0.0) ? 0.0 : part;
    double result = (is_nan(part) || part < dz_field) ? part : dz_field;
    part = zbot - top;
    part = (part < 0.0) ? 0.0 : part;
    result = (is_nan(part) || part < result) ? part : result;
    if (fraction) {
      const double f = (dz_field == 0.0) ? canonical_nan() : dz_field;
      const double g = (result == 0.0) ? canonical_nan() : result;
      result = g / f;
    }
    out[z * plane + i] = result;
  }
}

// synthetic field: lo + scale*u(splitmix64(seed ^ field<<60 ^ global_index))
template <typename TOut>
__global__ __launch_bounds__(kBlock) void k_synth(TOut* __restrict__ out, int64_t nt, int64_t nz,
                                                  int64_t ny, int64_t nx, int64_t t0, int64_t NY,
                                                  int64_t NX, int64_t y0, int64_t x0,
                                                  unsigned long long seed, int field_id, double lo,
                                                  double scale, const double* __restrict__ mask3d) {
  const int64_t n3 = nz * ny * nx;
  const int64_t n = nt * n3;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t t = i / n3;
    const int64_t r = i - t * n3;
    const int64_t z = r / (ny * nx);
    const int64_t r2 = r - z * ny * nx;
    const int64_t y = r2 / nx;
    const int64_t x = r2 - y * nx;
    const unsigned long long g =
        (unsigned long long)((((t0 + t) * nz + z) * NY + (y0 + y)) * NX + (x0 + x));
    const unsigned long long h = splitmix64(seed ^ ((unsigned long long)field_id << 60) ^ g);
    const double u = (double)(h >> 11) * 0x1.0p-53;
    double v = lo + scale * u;
    if (mask3d != nullptr && is_nan(mask3d[r])) v = canonical_nan();
    out[i] = (TOut)v;
  }
}

}  // namespace mlx

// =====================================================================================
// C ABI
// =====================================================================================
namespace {

using namespace mlx;

thread_local char g_err[512] = "";
// the instantiation the calling thread's last K1 / K2 dispatch launched (mlx_last_kernel)
thread_local char g_kernel[160] = "";

template <typename T>
const char* type_name() {
  return sizeof(T) == 8 ? "double" : "float";
}
inline const char* tf(bool b) { return b ? "true" : "false"; }

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int hip_status(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

}  // namespace

// the other translation units of the library (momlevel_promote.hip) report errors through the same
// per-thread buffer
int mlx::detail::fail(int code, const char* msg) { return ::fail(code, msg); }
int mlx::detail::hip_status(hipError_t e, const char* what) { return ::hip_status(e, what); }

namespace {

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

inline int64_t ceil_div(int64_t a, int64_t b) { return a / b + (a % b != 0); }  // a >= 0, b > 0

// overflow-checked product of non-negative extents; false when it does not fit (or tops `limit`)
inline bool mul_fits(int64_t a, int64_t b, int64_t* out, int64_t limit = INT64_MAX / 16) {
  return !__builtin_mul_overflow(a, b, out) && *out >= 0 && *out <= limit;
}

// total element count nt*nz*plane of a 4-D field must be addressable (and its byte size fit int64)
inline bool extents_fit(int64_t nt, int64_t nz, int64_t plane) {
  int64_t n3, n4;
  return mul_fits(nz, plane, &n3) && mul_fits(nt, n3, &n4);
}

// cells a K1 block covers within one z level
constexpr int kU64 = 4, kVec64 = 2;  // 8 cells/thread, 2048 cells/block (fast f64)
constexpr int kU32 = 2, kVec32 = 4;  // 8 cells/thread (fast f32)
// NB: every variant of a dtype (steric / thermosteric / halosteric / all-in-one, any nt) must
// share ONE tiling: the reference state's masso0 (VAR 0, nt=1) has to equal masso(t=0) of the
// held-field launches bit for bit, and the partial-sum order is a function of the tiling.
constexpr int kUGen = 4;             // generic: 4 scalar cells/thread, 1024 cells/block
constexpr int kTChunk = 32;          // K1 time steps per block (grid.z = ceil(nt / t_chunk)), steric
constexpr int kTChunkHeld = 64;      // held-field variants and the all-variants pass: vol0 and the
                                     // held field are re-read once per chunk -- at 8 B/cell that is
                                     // 10 % extra traffic with 32-step chunks; 64..120 steps measured
                                     // 1-3.5 % faster (profiles/r02_tune_tchunk.log), steric is flat
constexpr int kNTIGen = 8;           // generic scalar path

// Columns and time steps per thread of the single-variant fast K2 kernels, chosen PER INSTANTIATION
// (round 4): by field type, variant, and whether delta_rho is stored.  More steps amortise rho0m /
// dz / the held field over more cells and keep more bytes in flight per wave; fewer steps -- and
// fewer columns -- need fewer registers (column sums + loads in flight) and run more waves per SIMD.
// Never changes a result: every (time step, column) sum adds its levels in the same order.
// Measured on one box, one process per library (scripts/ab_k2.py, profiles/r04_tune_k2_nti_*.log,
// profiles/r04_tune_k2_f32_columns.log).  Tuning builds override the tables:
// -DMLX_TUNE_NTI64=n / -DMLX_TUNE_NTI32=n / -DMLX_TUNE_K2_VEC32=n.
#ifndef MLX_TUNE_NTI64
#define MLX_TUNE_NTI64 0
#endif
#ifndef MLX_TUNE_NTI32
#define MLX_TUNE_NTI32 0
#endif
#ifndef MLX_TUNE_K2_VEC32
#define MLX_TUNE_K2_VEC32 0
#endif
#ifndef MLX_TUNE_K2_TBMAJOR
#define MLX_TUNE_K2_TBMAJOR 1
#endif
#ifndef MLX_TUNE_K2_LDSACC
#define MLX_TUNE_K2_LDSACC 0
#endif
// float32 fields: FOUR columns per thread (one 16-byte load per field, level and step) only for the
// steric pass without delta_rho; every other pass runs TWO (8-byte loads): half the registers per
// step buy twice the steps in flight at the same occupancy -- held-field passes 7-8 % faster without
// delta_rho, 6-18 % with it, the steric pass with delta_rho 4-12 %; the steric eta-only pass, which
// streams two fields per cell and stores next to nothing, is 3-5 % slower that way and stays at four.
constexpr int k2_vec32(int var, bool drho) {
  if (MLX_TUNE_K2_VEC32) return MLX_TUNE_K2_VEC32;
  return (var == kVarSteric && !drho) ? 4 : 2;
}
// float64 fields: two columns per thread (one 16-byte load per field, level and step), every pass.
// (Round 4 ran the held-field passes WITH delta_rho -- what thermosteric(ds) / halosteric(ds) run by
// default -- on ONE column and 32 steps: 3-13 % faster while every time block re-read rho0m and the
// held slab from HBM.  With a tile's time blocks side by side on one XCD (k_steric_local, round 5)
// those re-reads are L2 hits and the trade reverses: two columns x 8..16 steps and one column x
// 6..24 steps all measure 15.2-15.8 ms where one column x 32 steps takes 17.2 -- profiles/
// r05_tune_k2_held_drho.log, r05_tune_k2_nti_tbmajor.log.)
#ifndef MLX_TUNE_K2_VEC64
#define MLX_TUNE_K2_VEC64 0
#endif
constexpr int k2_vec64(int var, bool drho) {
  if (MLX_TUNE_K2_VEC64) return MLX_TUNE_K2_VEC64;
  return 2;
}
constexpr int k2_nti(bool f64, int var, bool drho) {
  if (f64 && MLX_TUNE_NTI64) return MLX_TUNE_NTI64;
  if (!f64 && MLX_TUNE_NTI32) return MLX_TUNE_NTI32;
  // float64 (two columns): 16 steps for the steric pass (8 and 12 measured 0-4 % slower in round 4,
  // 8 within 2.5 % either way in round 5); 8 for the held-field passes, with and without delta_rho
  // -- ONE instantiation per variant: half the streamed bytes of the steric pass per cell, so
  // occupancy weighs more than amortising the per-level work, and since round 5 a short time block
  // no longer costs HBM re-reads of rho0m and the held slab (its siblings share them through L2):
  // eta-only 8 vs 12 steps 2-3 % faster, with delta_rho 8 / 12 / 16 equal within noise.
  // float32, four columns (steric, eta only): 6 steps (3 waves per SIMD; 4 and 8 measured 3-10 %
  // slower).  float32, two columns: 12 steps for the held-field passes without delta_rho, 16 for
  // everything that stores delta_rho (8 / 10 / 12 / 16 swept in round 4, 6 / 8 / 16 / 24 again in
  // round 5: no shape beats them beyond the box's 8 % run-to-run scatter on these passes).
  if (f64) return (var == kVarSteric) ? 16 : 8;
  if (var == kVarSteric && !drho) return 6;
  return drho ? 16 : 12;
}

constexpr int kKnownFlags = MLX_FLAG_SKIP_DRY | MLX_FLAG_FMA | MLX_FLAG_TCHUNK_MASK;

inline bool mixed_dtype(int dtype) {
  return dtype == MLX_DTYPE_T32_S64 || dtype == MLX_DTYPE_T64_S32;
}
// element size of theta (is_T) / salinity in memory
inline size_t elem_size(int dtype, bool is_T) {
  if (dtype == MLX_DTYPE_F64) return 8;
  if (dtype == MLX_DTYPE_T32_S64) return is_T ? 4 : 8;
  if (dtype == MLX_DTYPE_T64_S32) return is_T ? 8 : 4;
  return 4;
}

// `mixed_ok`: the steric kernels (K1 / K2) take theta and salinity of different dtypes; the
// pointwise maps do not (mlx_eos_map_promote covers every combination there)
int check_dtype(int dtype, bool mixed_ok) {
  if (mixed_dtype(dtype)) {
    if (!mixed_ok)
      return fail(MLX_E_ENUM, "theta/salinity of different dtypes: use mlx_eos_map_promote");
    return 0;
  }
  if (dtype != MLX_DTYPE_F64 && dtype != MLX_DTYPE_F32 && dtype != MLX_DTYPE_F32_UPCAST)
    return fail(MLX_E_ENUM, "dtype must be one of MLX_DTYPE_*");
  return 0;
}

int check_common(const void* T, const void* S, int dtype, const double* p, int p_mode, int eos,
                 int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS,
                 bool mixed_ok = false) {
  if (!T || !S) return fail(MLX_E_NULL, "T and S must not be NULL");
  if (int rc = check_dtype(dtype, mixed_ok)) return rc;
  if (eos != MLX_EOS_WRIGHT && eos != MLX_EOS_LINEAR) return fail(MLX_E_ENUM, "unknown eos");
  if (p_mode < MLX_P_SCALAR || p_mode > MLX_P_FULL4D) return fail(MLX_E_ENUM, "unknown p_mode");
  if (!p && eos == MLX_EOS_WRIGHT) return fail(MLX_E_NULL, "p must not be NULL for the Wright EOS");
  // a NULL p (linear EOS) is replaced by a placeholder pointer the kernels never dereference -- which
  // holds for MLX_P_SCALAR only: the array modes would index the placeholder
  if (!p && p_mode != MLX_P_SCALAR) return fail(MLX_E_NULL, "a NULL p requires p_mode MLX_P_SCALAR");
  if (nt <= 0 || nz <= 0 || plane <= 0) return fail(MLX_E_SHAPE, "nt, nz, plane must be > 0");
  if (nz > 65535) return fail(MLX_E_SHAPE, "nz must be <= 65535");
  if (nt > 2147483647LL) return fail(MLX_E_SHAPE, "nt too large");
  if (plane > (int64_t)1 << 40) return fail(MLX_E_SHAPE, "plane too large");
  if (!extents_fit(nt, nz, plane)) return fail(MLX_E_SHAPE, "nt*nz*plane overflows");
  if (sT < 0 || sS < 0) return fail(MLX_E_SHAPE, "time strides must be >= 0");
  {  // the last time step must be addressable: (nt-1)*stride + nz*plane
    int64_t span;
    if (!mul_fits(nt - 1, sT > sS ? sT : sS, &span)) return fail(MLX_E_SHAPE, "time stride too large");
  }
  if (!aligned(T, elem_size(dtype, true)) || !aligned(S, elem_size(dtype, false)))
    return fail(MLX_E_ALIGN, "T/S not element-aligned");
  if (p && !aligned(p, 8)) return fail(MLX_E_ALIGN, "p not 8-byte aligned");
  return 0;
}

// cells per pack of the fast kernels: 2 when a float64 field takes part (also theta and salinity of
// different dtypes: the float32 field then comes as 8-byte loads), 4 for float32 fields
inline int vec_of(int dtype) { return (dtype == MLX_DTYPE_F64 || mixed_dtype(dtype)) ? kVec64 : kVec32; }

// the dwordx4 kernels need: Wright EOS, z-profile pressure, whole packs per plane and per time
// stride, 16-byte aligned operands; everything else takes the generic (scalar) twin
// (p_mode MLX_P_FULL3D -- `patm` as a (yh,xh) DataArray -- qualifies too where the caller passes
// the pressure pointer as `p3d`: the fast kernels then read the pressure FIELD with the same
// 16-byte loads, template argument P3D; a z profile needs no alignment beyond its elements')
bool fast_layout(int dtype, int p_mode, int eos, int64_t plane, int64_t sT, int64_t sS,
                 std::initializer_list<const void*> ptrs, const void* p3d = nullptr) {
  const int vec = vec_of(dtype);
  if (eos != MLX_EOS_WRIGHT) return false;
  if (p_mode != MLX_P_ZPROF && !(p3d && p_mode == MLX_P_FULL3D && aligned(p3d, 16))) return false;
  if (plane % vec || sT % vec || sS % vec) return false;
  for (const void* q : ptrs)
    if (q && !aligned(q, 16)) return false;
  return true;
}

// ---- K1 dispatch -------------------------------------------------------------------------
struct K1Args {
  dim3 grid;
  hipStream_t st;
  const void *T, *S, *T0, *S0;
  const double *vol0, *p;
  int p_mode, eos, nt, t_chunk;
  int64_t plane, sT, sS;
  double* partials;
  int64_t nblk;
  bool p3d;  // fast kernels: the pressure is a (z,y,x) field (template argument P3D)
};

template <typename TIn, int VEC, int U, int VAR, int MODE, bool GEN, bool SKIP, bool FMA,
          bool P3D>
void k1_launch(const K1Args& a) {
  snprintf(g_kernel, sizeof(g_kernel), "k_steric_global<%s,%d,%d,%d,%d,%s,%s,%s%s>",
           type_name<TIn>(), VEC, U, VAR, MODE, tf(GEN), tf(SKIP), tf(FMA), P3D ? ",true" : "");
  hipLaunchKernelGGL((k_steric_global<TIn, VEC, U, VAR, MODE, GEN, SKIP, FMA, P3D>), a.grid,
                     dim3(kBlock), 0, a.st, (const TIn*)a.T, (const TIn*)a.S, (const TIn*)a.T0,
                     (const TIn*)a.S0, a.vol0, a.p, a.p_mode, a.eos, a.nt, a.t_chunk, a.plane,
                     a.sT, a.sS, a.partials, a.nblk);
}

template <typename TIn, int VEC, int U, int VAR, int MODE, bool GEN, bool SKIP, bool FMA>
void k1_go(const K1Args& a) {
  if constexpr (!GEN && VAR != kVarAll) {
    if (a.p3d) {
      k1_launch<TIn, VEC, U, VAR, MODE, GEN, SKIP, FMA, true>(a);
      return;
    }
  }
  k1_launch<TIn, VEC, U, VAR, MODE, GEN, SKIP, FMA, false>(a);
}

template <typename TIn, int VEC, int U, int VAR, int MODE, bool GEN>
void k1_flags(const K1Args& a, bool skip, bool fma) {
  constexpr int FM = MODE;   // (faithful float32 keeps its float32 polynomial under MLX_FLAG_FMA)
  constexpr bool S1 = !GEN;  // the generic twin has no skipping instantiation
  if constexpr (IsMixed<MODE>::value) {  // exact arithmetic only (the entry points refuse the flag)
    if (skip && S1) k1_go<TIn, VEC, U, VAR, MODE, GEN, S1, false>(a);
    else k1_go<TIn, VEC, U, VAR, MODE, GEN, false, false>(a);
  } else if (fma) {
    if (skip && S1) k1_go<TIn, VEC, U, VAR, FM, GEN, S1, true>(a);
    else k1_go<TIn, VEC, U, VAR, FM, GEN, false, true>(a);
  } else {
    if (skip && S1) k1_go<TIn, VEC, U, VAR, MODE, GEN, S1, false>(a);
    else k1_go<TIn, VEC, U, VAR, MODE, GEN, false, false>(a);
  }
}

template <typename TIn, int VEC, int U, int MODE, bool GEN>
void k1_var(const K1Args& a, int var, bool skip, bool fma) {
  if constexpr (GEN) {  // held fields reach the generic twin as stride-0 streams (VAR 0)
    if (var == kVarAll) k1_flags<TIn, VEC, U, kVarAll, MODE, GEN>(a, skip, fma);
    else k1_flags<TIn, VEC, U, kVarSteric, MODE, GEN>(a, skip, fma);
  } else {
    switch (var) {
      case kVarSteric: k1_flags<TIn, VEC, U, kVarSteric, MODE, GEN>(a, skip, fma); break;
      case kVarHalo: k1_flags<TIn, VEC, U, kVarHalo, MODE, GEN>(a, skip, fma); break;
      case kVarThermo: k1_flags<TIn, VEC, U, kVarThermo, MODE, GEN>(a, skip, fma); break;
      default: k1_flags<TIn, VEC, U, kVarAll, MODE, GEN>(a, skip, fma); break;
    }
  }
}

void k1_dispatch(const K1Args& a, int dtype, bool fast, int var, bool skip, bool fma) {
  if (fast) {
    if (dtype == MLX_DTYPE_F64) k1_var<double, kVec64, kU64, kF64, false>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_F32) k1_var<float, kVec32, kU32, kF32Faithful, false>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_T32_S64) k1_var<double, kVec64, kU64, kMixT32, false>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_T64_S32) k1_var<double, kVec64, kU64, kMixS32, false>(a, var, skip, fma);
    else k1_var<float, kVec32, kU32, kF32Upcast, false>(a, var, skip, fma);
  } else {
    if (dtype == MLX_DTYPE_F64) k1_var<double, 1, kUGen, kF64, true>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_F32) k1_var<float, 1, kUGen, kF32Faithful, true>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_T32_S64) k1_var<double, 1, kUGen, kMixT32, true>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_T64_S32) k1_var<double, 1, kUGen, kMixS32, true>(a, var, skip, fma);
    else k1_var<float, 1, kUGen, kF32Upcast, true>(a, var, skip, fma);
  }
}

int64_t k1_blocks(bool fast, int dtype, int64_t plane) {
  const int64_t cells = fast ? (int64_t)kBlock * 8 : (int64_t)kBlock * kUGen;
  (void)dtype;  // 8 cells per thread for both fast dtypes
  return ceil_div(plane, cells);
}

int k1_time_chunk(int flags, int64_t nt, int var) {
  const int hint = (flags & MLX_FLAG_TCHUNK_MASK) >> 8;
  int64_t tc = hint ? (int64_t)hint * 8 : (var == kVarSteric ? kTChunk : kTChunkHeld);
  if (tc > nt) tc = nt;
  return (int)tc;
}

// shared body of mlx_steric_global (nout 1) and mlx_steric_global_decomp (nout 4)
int steric_global_impl(const void* T, const void* S, const void* T0, const void* S0, int var,
                       int dtype, const double* vol0, const double* p, int p_mode, int eos,
                       int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS, int flags,
                       double* out, void* workspace, size_t workspace_bytes, void* stream,
                       const char* name) {
  const int nout = (var == kVarAll) ? 4 : 1;
  if (flags & ~kKnownFlags) return fail(MLX_E_ENUM, "unknown flag bits");
  if (int rc = check_common(T, S, dtype, p, p_mode, eos, nt, nz, plane, sT, sS, true)) return rc;
  if (mixed_dtype(dtype) && (flags & MLX_FLAG_FMA))
    return fail(MLX_E_ENUM, "MLX_FLAG_FMA is not available for theta/salinity of different dtypes");
  if (!vol0 || !out) return fail(MLX_E_NULL, "vol0 and the output must not be NULL");
  if (!aligned(vol0, 8) || !aligned(out, 8)) return fail(MLX_E_ALIGN, "vol0/out not 8-byte aligned");
  if (var == kVarAll) {
    if (!T0 || !S0) return fail(MLX_E_NULL, "T0 and S0 must not be NULL");
    if (!aligned(T0, elem_size(dtype, true)) || !aligned(S0, elem_size(dtype, false)))
      return fail(MLX_E_ALIGN, "T0/S0 not element-aligned");
  }
  if (!workspace) return fail(MLX_E_NULL, "workspace must not be NULL");
  const size_t need = (size_t)nout * mlx_steric_global_workspace_bytes(nt, nz, plane);
  if (!aligned(workspace, 8) || workspace_bytes < need)
    return fail(MLX_E_WORKSPACE, "workspace smaller than the *_workspace_bytes() query");
  const bool skip = (flags & MLX_FLAG_SKIP_DRY) != 0, fma = (flags & MLX_FLAG_FMA) != 0;
  // (the all-variants kernel has no registers left for a pressure field: generic twin)
  bool fast = fast_layout(dtype, p_mode, eos, plane, sT, sS, {T, S, T0, S0, vol0},
                          var != kVarAll ? p : nullptr);
  if (var != kVarAll) {
    if (sT == 0 && sS == 0) fast = false;  // both held: nothing streams; the generic twin reloads
    else if (sT == 0) { var = kVarHalo; T0 = T; }
    else if (sS == 0) { var = kVarThermo; S0 = S; }
  }
  K1Args a;
  a.p3d = fast && p_mode == MLX_P_FULL3D;
  const int64_t gx = k1_blocks(fast, dtype, plane);
  a.t_chunk = k1_time_chunk(flags, nt, var);
  if (ceil_div(nt, a.t_chunk) > 65535) return fail(MLX_E_SHAPE, "nt too large for one call: chunk it");
  if (gx > 2147483647LL) return fail(MLX_E_SHAPE, "plane too large");
  a.grid = dim3((unsigned)gx, (unsigned)nz, (unsigned)ceil_div(nt, a.t_chunk));
  a.st = (hipStream_t)stream;
  a.T = T; a.S = S; a.T0 = T0 ? T0 : T; a.S0 = S0 ? S0 : S;
  a.vol0 = vol0; a.p = p ? p : vol0;  // p is never dereferenced for the linear EOS
  a.p_mode = p_mode; a.eos = eos; a.nt = (int)nt;
  a.plane = plane; a.sT = sT; a.sS = sS;
  a.partials = (double*)workspace; a.nblk = gx * nz;
  k1_dispatch(a, dtype, fast, var, skip, fma);
  if (int rc = hip_status(hipGetLastError(), name)) return rc;
  hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)(nout * nt)), dim3(kBlock), 0, a.st,
                     a.partials, a.nblk, out);
  return hip_status(hipGetLastError(), "k_reduce_rows launch");
}

// ---- K2 dispatch -------------------------------------------------------------------------
struct K2Args {
  dim3 grid;
  hipStream_t st;
  const void *T, *S, *T0, *S0;
  const double *rho0m, *surf, *dz, *z_i, *deptho, *p;
  int p_mode, eos, nt, nz;
  double neg_inv_rhozero;
  int64_t plane, sT, sS;
  double *drho, *eta;
  int64_t drho_vstride, eta_vstride;
  bool p3d;  // fast kernels: the pressure is a (z,y,x) field (template argument P3D)
  int ntb_major;  // > 0: 1-D grid, a tile's time blocks adjacent on one XCD (k_steric_local)
};

template <typename TIn, int VEC, int NTI, int VAR, int MODE, bool GEN, bool SKIP, bool FMA,
          bool P3D, bool LDSACC = false>
void k2_launch(const K2Args& a) {
  snprintf(g_kernel, sizeof(g_kernel), "k_steric_local<%s,%d,%d,%d,%d,%s,%s,%s%s%s>",
           type_name<TIn>(), VEC, NTI, VAR, MODE, tf(GEN), tf(SKIP), tf(FMA),
           (P3D || LDSACC) ? (P3D ? ",true" : ",false") : "", LDSACC ? ",lds" : "");
  hipLaunchKernelGGL((k_steric_local<TIn, VEC, NTI, VAR, MODE, GEN, SKIP, FMA, P3D, LDSACC>), a.grid,
                     dim3(kBlock), 0, a.st, (const TIn*)a.T, (const TIn*)a.S, (const TIn*)a.T0,
                     (const TIn*)a.S0, a.rho0m, a.surf, a.dz, a.z_i, a.deptho, a.p, a.p_mode, a.eos,
                     a.neg_inv_rhozero, a.nt, a.nz, a.plane, a.sT, a.sS, a.drho, a.drho_vstride,
                     a.eta, a.eta_vstride, a.ntb_major);
}

template <typename TIn, int VEC, int NTI, int VAR, int MODE, bool GEN, bool SKIP, bool FMA>
void k2_go(const K2Args& a) {
  if constexpr (!GEN && VAR != kVarAll) {
    if (a.p3d) {
      k2_launch<TIn, VEC, NTI, VAR, MODE, GEN, SKIP, FMA, true>(a);
      return;
    }
  }
  k2_launch<TIn, VEC, NTI, VAR, MODE, GEN, SKIP, FMA, false>(a);
}

template <typename TIn, int VEC, int NTI, int VAR, int MODE, bool GEN>
void k2_flags(const K2Args& a, bool skip, bool fma) {
  constexpr int FM = MODE;
  constexpr bool S1 = !GEN;
  if constexpr (IsMixed<MODE>::value) {  // exact arithmetic only
    if (skip && S1) k2_go<TIn, VEC, NTI, VAR, MODE, GEN, S1, false>(a);
    else k2_go<TIn, VEC, NTI, VAR, MODE, GEN, false, false>(a);
  } else if (fma) {
    if (skip && S1) k2_go<TIn, VEC, NTI, VAR, FM, GEN, S1, true>(a);
    else k2_go<TIn, VEC, NTI, VAR, FM, GEN, false, true>(a);
  } else {
    if (skip && S1) k2_go<TIn, VEC, NTI, VAR, MODE, GEN, S1, false>(a);
    else k2_go<TIn, VEC, NTI, VAR, MODE, GEN, false, false>(a);
  }
}

// a single-variant fast kernel with the time steps per thread of its instantiation (k2_nti)
template <typename TIn, int VEC, int VAR, int MODE>
void k2_single(const K2Args& a, bool skip, bool fma) {
  constexpr bool F64 = sizeof(TIn) == 8;
  // (the columns per thread are the instantiation's too: k2_vec32, k2_vec64; theta / S of different
  // dtypes keep the two columns of their float64 shape)
  constexpr int V1 = !F64 ? k2_vec32(VAR, true) : (MODE == kF64 ? k2_vec64(VAR, true) : VEC);
  constexpr int V0 = !F64 ? k2_vec32(VAR, false) : (MODE == kF64 ? k2_vec64(VAR, false) : VEC);
  if constexpr (MLX_TUNE_K2_LDSACC && VAR != kVarSteric && (MODE == kF64 || MODE == kF32Faithful)) {
    // the experiment: eta-only held-field passes, exact arithmetic, no dry skipping, z-profile
    // pressure: 32 steps of one column (float64) / two columns (float32) per thread, sums in LDS
    if (a.drho == nullptr && !skip && !fma && !a.p3d) {
      k2_launch<TIn, F64 ? 1 : 2, F64 ? 32 : 16, VAR, MODE, false, false, false, false, true>(a);
      return;
    }
  }
  if (a.drho != nullptr) k2_flags<TIn, V1, k2_nti(F64, VAR, true), VAR, MODE, false>(a, skip, fma);
  else k2_flags<TIn, V0, k2_nti(F64, VAR, false), VAR, MODE, false>(a, skip, fma);
}

// NTI1: time steps per thread of the generic single-variant kernel (the fast ones: k2_nti);
// VEC3/NTI3: columns and time steps per thread of the all-variants kernel (three sets of column
// sums in registers)
template <typename TIn, int VEC, int NTI1, int VEC3, int NTI3, int MODE, bool GEN>
void k2_var(const K2Args& a, int var, bool skip, bool fma) {
  if (var == kVarAll) {
    k2_flags<TIn, VEC3, NTI3, kVarAll, MODE, GEN>(a, skip, fma);
  } else if constexpr (GEN) {  // held fields reach the generic twin as stride-0 streams
    k2_flags<TIn, VEC, NTI1, kVarSteric, MODE, GEN>(a, skip, fma);
  } else {
    if (var == kVarSteric) k2_single<TIn, VEC, kVarSteric, MODE>(a, skip, fma);
    else if (var == kVarHalo) k2_single<TIn, VEC, kVarHalo, MODE>(a, skip, fma);
    else k2_single<TIn, VEC, kVarThermo, MODE>(a, skip, fma);
  }
}

// all-variants kernel: 2 columns x 8 time steps per thread for BOTH dtypes (float32 theta/S then
// come as float2 = 8-byte loads; four columns' worth of held parts and three sets of column sums
// would not fit 256 VGPRs)
constexpr int kNTI64All = 8, kNTI32All = 8, kNTIGenAll = 4, kVec32All = 2;

// shared body of mlx_steric_local (var 0/1/2 from the strides) and mlx_steric_local_decomp (var 3)
int steric_local_impl(const void* T, const void* S, const void* T0, const void* S0, int var,
                      int dtype, const double* rho0m, const double* vol0_surface, const double* dz,
                      const double* z_i, const double* deptho, const double* p, int p_mode, int eos,
                      double neg_inv_rhozero, int64_t nt, int64_t nz, int64_t plane, int64_t sT,
                      int64_t sS, int flags, double* delta_rho_out, int64_t drho_vstride,
                      double* eta_out, int64_t eta_vstride, void* stream) {
  if (flags & ~(MLX_FLAG_SKIP_DRY | MLX_FLAG_FMA)) return fail(MLX_E_ENUM, "unknown flag bits");
  const bool skip = (flags & MLX_FLAG_SKIP_DRY) != 0, fma = (flags & MLX_FLAG_FMA) != 0;
  if (int rc = check_common(T, S, dtype, p, p_mode, eos, nt, nz, plane, sT, sS, true)) return rc;
  if (mixed_dtype(dtype) && fma)
    return fail(MLX_E_ENUM, "MLX_FLAG_FMA is not available for theta/salinity of different dtypes");
  if (!rho0m || !vol0_surface || !eta_out)
    return fail(MLX_E_NULL, "rho0m, vol0_surface and eta_out must not be NULL");
  if (!dz && (!z_i || !deptho))
    return fail(MLX_E_NULL, "either dz or both z_i and deptho must be given");
  for (const void* q : {(const void*)rho0m, (const void*)vol0_surface, (const void*)dz,
                        (const void*)z_i, (const void*)deptho, (const void*)delta_rho_out,
                        (const void*)eta_out})
    if (q && !aligned(q, 8)) return fail(MLX_E_ALIGN, "operands not 8-byte aligned");
  if (var == kVarAll) {
    if (!T0 || !S0) return fail(MLX_E_NULL, "T0 and S0 must not be NULL");
    if (!aligned(T0, elem_size(dtype, true)) || !aligned(S0, elem_size(dtype, false)))
      return fail(MLX_E_ALIGN, "T0/S0 not element-aligned");
    int64_t n3, n4;
    if (!mul_fits(nz, plane, &n3) || !mul_fits(nt, n3, &n4)) return fail(MLX_E_SHAPE, "overflow");
    if (eta_vstride < nt * plane || (delta_rho_out && drho_vstride < n4))
      return fail(MLX_E_SHAPE, "variant strides must be >= the size of one variant's field");
  }
  const bool f64 = (dtype == MLX_DTYPE_F64) || mixed_dtype(dtype);  // the float64 kernel SHAPE
  // both strides must be whole packs too in the all-variants kernel (three fields, one base)
  bool fast = fast_layout(dtype, p_mode, eos, plane, sT, sS,
                          {T, S, T0, S0, rho0m, vol0_surface, eta_out, dz, dz ? nullptr : deptho,
                           delta_rho_out},
                          var != kVarAll ? p : nullptr) &&
              !(sT == 0 && sS == 0);
  if (var == kVarAll && fast) {
    const int vec = vec_of(dtype);
    if (eta_vstride % vec || drho_vstride % vec) fast = false;
  }
  if (var != kVarAll) var = (sT == 0) ? kVarHalo : ((sS == 0) ? kVarThermo : kVarSteric);
  // (float32 theta/S evaluated in float64 -- MLX_DTYPE_F32_UPCAST, or MLX_FLAG_FMA on float32 input
  //  -- were tried in the float64 kernel's shape, 2 columns x 16 steps with float2 loads, because
  //  four columns of float64 polynomial need 254 VGPRs: 64.6 ms instead of 43 for the upcast pass
  //  at the roofline grid, no change for the fused one -- 8-byte loads stream worse than they save.
  //  Neither is a default path.)
  const bool f32_fields = dtype == MLX_DTYPE_F32 || dtype == MLX_DTYPE_F32_UPCAST;
  int v = !fast ? 1
                : (var == kVarAll && !f64) ? kVec32All
                : f32_fields ? k2_vec32(var, delta_rho_out != nullptr)
                : (dtype == MLX_DTYPE_F64 && var != kVarAll) ? k2_vec64(var, delta_rho_out != nullptr)
                             : vec_of(dtype);
  int nti = (var == kVarAll) ? (fast ? (f64 ? kNTI64All : kNTI32All) : kNTIGenAll)
                             : (fast ? k2_nti(f64, var, delta_rho_out != nullptr) : kNTIGen);
  if (MLX_TUNE_K2_LDSACC && fast && (var == kVarHalo || var == kVarThermo) && !delta_rho_out &&
      !skip && !fma && p_mode != MLX_P_FULL3D &&
      (dtype == MLX_DTYPE_F64 || dtype == MLX_DTYPE_F32)) {  // (the experiment's shape: k2_single)
    v = (dtype == MLX_DTYPE_F64) ? 1 : 2;
    nti = (dtype == MLX_DTYPE_F64) ? 32 : 16;
  }
  if (ceil_div(nt, nti) > 65535) return fail(MLX_E_SHAPE, "nt too large for one call: chunk it");
  const int64_t gx = ceil_div(plane, (int64_t)kBlock * v);
  if (gx > 2147483647LL) return fail(MLX_E_SHAPE, "plane too large");
  K2Args a;
  a.grid = dim3((unsigned)gx, (unsigned)ceil_div(nt, nti));
  a.ntb_major = 0;
  {  // time blocks of a tile side by side on one XCD (see the kernel) when there is more than one
    const int64_t ntb = ceil_div(nt, nti);
    const int64_t blocks = ceil_div(gx, 8) * 8 * ntb;
    // (a HIP grid holds fewer than 2^32 THREADS: beyond 2^24 - 1 blocks of kBlock the 1-D form
    //  would not launch where the (tiles, time blocks) grid does -- ADVICE r5)
    if (MLX_TUNE_K2_TBMAJOR && ntb > 1 && blocks <= kK2MaxBlocks1D) {
      a.grid = dim3((unsigned)blocks);
      a.ntb_major = (int)ntb;
    }
  }
  a.st = (hipStream_t)stream;
  a.T = T; a.S = S; a.T0 = T0 ? T0 : T; a.S0 = S0 ? S0 : S;
  a.rho0m = rho0m; a.surf = vol0_surface; a.dz = dz; a.z_i = z_i;
  a.deptho = deptho; a.p = p ? p : rho0m; a.p_mode = p_mode; a.eos = eos; a.nt = (int)nt;
  a.nz = (int)nz; a.neg_inv_rhozero = neg_inv_rhozero; a.plane = plane; a.sT = sT; a.sS = sS;
  a.drho = delta_rho_out; a.eta = eta_out; a.drho_vstride = drho_vstride; a.eta_vstride = eta_vstride;
  a.p3d = fast && p_mode == MLX_P_FULL3D;
  if (fast) {
    if (dtype == MLX_DTYPE_T32_S64)
      k2_var<double, kVec64, 0, kVec64, kNTI64All, kMixT32, false>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_T64_S32)
      k2_var<double, kVec64, 0, kVec64, kNTI64All, kMixS32, false>(a, var, skip, fma);
    else if (f64) k2_var<double, kVec64, 0, kVec64, kNTI64All, kF64, false>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_F32)
      k2_var<float, kVec32, 0, kVec32All, kNTI32All, kF32Faithful, false>(a, var, skip, fma);
    else k2_var<float, kVec32, 0, kVec32All, kNTI32All, kF32Upcast, false>(a, var, skip, fma);
  } else {
    if (dtype == MLX_DTYPE_F64) k2_var<double, 1, kNTIGen, 1, kNTIGenAll, kF64, true>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_F32)
      k2_var<float, 1, kNTIGen, 1, kNTIGenAll, kF32Faithful, true>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_T32_S64)
      k2_var<double, 1, kNTIGen, 1, kNTIGenAll, kMixT32, true>(a, var, skip, fma);
    else if (dtype == MLX_DTYPE_T64_S32)
      k2_var<double, 1, kNTIGen, 1, kNTIGenAll, kMixS32, true>(a, var, skip, fma);
    else k2_var<float, 1, kNTIGen, 1, kNTIGenAll, kF32Upcast, true>(a, var, skip, fma);
  }
  return hip_status(hipGetLastError(), "k_steric_local launch");
}

}  // namespace

// (U, nontemporal store) per mix: the fastest shape of scripts/tune_probe.hip's sweep on hashed
// data -- profiles/r05_tune_probe_hashed.log
template <typename TIn, int NIN, bool WRITE, int U, bool NTS, int VEC = 16 / sizeof(TIn)>
static int launch_probe_mix(const void* a, const void* b, int64_t n, double* out, hipStream_t st) {
  const int64_t npacks = n / VEC;
  const int64_t ntiles = ceil_div(npacks, (int64_t)kBlock * U);
  const dim3 grid((unsigned)(ntiles < kProbeMaxBlocks ? ntiles : kProbeMaxBlocks));
  hipLaunchKernelGGL((k_stream_probe_mix<TIn, NIN, WRITE, U, NTS, VEC>), grid, dim3(kBlock), 0, st,
                     (const TIn*)a, (const TIn*)b, npacks, out);
  return hip_status(hipGetLastError(), "k_stream_probe_mix launch");
}

extern "C" {

int mlx_version(void) { return MLX_ABI_VERSION; }

int mlx_last_error(char* buf, size_t n) {
  if (buf && n) {
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(g_err);
}

int mlx_build_kind(void) { return MLX_BUILD_HIP; }

int mlx_last_kernel(char* buf, size_t n) {
  if (buf && n) {
    strncpy(buf, g_kernel, n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(g_kernel);
}

// ---------------------------------------------------------------------------- K0
static int eos_map_impl(const void* T, const void* S, int dtype, const double* p, int p_mode,
                        int eos, int func, double aux, int64_t nt, int64_t nz, int64_t plane,
                        int64_t sT, int64_t sS, int flags, double* out, void* stream) {
  if (flags & ~MLX_FLAG_FMA) return fail(MLX_E_ENUM, "mlx_eos_map takes MLX_FLAG_FMA only");
  if (int rc = check_common(T, S, dtype, p, p_mode, eos, nt, nz, plane, sT, sS)) return rc;
  if (!out) return fail(MLX_E_NULL, "out must not be NULL");
  if (!aligned(out, 8)) return fail(MLX_E_ALIGN, "out not 8-byte aligned");
  if (func < MLX_FUNC_DENSITY || func > MLX_FUNC_IBH) return fail(MLX_E_ENUM, "unknown func");
  if (func == MLX_FUNC_IBH && !p) return fail(MLX_E_NULL, "p must not be NULL");
  const bool fma = (flags & MLX_FLAG_FMA) != 0;
  if (fma && func != MLX_FUNC_DENSITY)
    return fail(MLX_E_ENUM, "MLX_FLAG_FMA applies to the density only");
  hipStream_t st = (hipStream_t)stream;
  const bool f64 = (dtype == MLX_DTYPE_F64);
  const int vec = vec_of(dtype);
  const bool fast = fast_layout(dtype, p_mode, eos, plane, sT, sS, {T, S, out},
                                func == MLX_FUNC_DENSITY ? p : nullptr) &&
                    (f64 || func == MLX_FUNC_DENSITY) && func != MLX_FUNC_IBH;
  const bool p3d = fast && p_mode == MLX_P_FULL3D;
  const double* pp = p ? p : out;  // never dereferenced for the linear EOS
  for (int64_t tb = 0; tb < nt; tb += 32768) {
    const int64_t ntc = (nt - tb < 32768) ? (nt - tb) : 32768;
    if (fast) {
      constexpr int U = 2;
      dim3 grid((unsigned)ceil_div(plane, (int64_t)kBlock * vec * U), (unsigned)nz, (unsigned)ntc);
#define MLX_LAUNCH_K0P(TIN, VEC, MODE, FUNC, FMA, P3D)                                           \
  hipLaunchKernelGGL((k_eos_map<TIN, VEC, U, MODE, FUNC, false, FMA, P3D>), grid, dim3(kBlock), \
                     0, st, (const TIN*)T, (const TIN*)S, pp, p_mode, eos, func, nz, plane, sT,  \
                     sS, tb, aux, out)
// (the density map takes a (z,y,x) pressure field on the fast path too: P3D)
#define MLX_LAUNCH_K0(TIN, VEC, MODE, FUNC, FMA)                                                 \
  do {                                                                                           \
    if constexpr (FUNC == kDensity) {                                                            \
      if (p3d) { MLX_LAUNCH_K0P(TIN, VEC, MODE, FUNC, FMA, true); break; }                       \
    }                                                                                            \
    MLX_LAUNCH_K0P(TIN, VEC, MODE, FUNC, FMA, false);                                            \
  } while (0)
      if (f64) {
        switch (func) {
          case MLX_FUNC_DENSITY:
            if (fma) MLX_LAUNCH_K0(double, 2, kF64, kDensity, true);
            else MLX_LAUNCH_K0(double, 2, kF64, kDensity, false);
            break;
          case MLX_FUNC_DRHO_DTEMP: MLX_LAUNCH_K0(double, 2, kF64, kDrhoDtemp, false); break;
          case MLX_FUNC_DRHO_DSAL: MLX_LAUNCH_K0(double, 2, kF64, kDrhoDsal, false); break;
          case MLX_FUNC_ALPHA: MLX_LAUNCH_K0(double, 2, kF64, kAlpha, false); break;
          default: MLX_LAUNCH_K0(double, 2, kF64, kBeta, false); break;
        }
      } else if (fma) {
        if (dtype == MLX_DTYPE_F32) MLX_LAUNCH_K0(float, 4, kF32Faithful, kDensity, true);
        else MLX_LAUNCH_K0(float, 4, kF32Upcast, kDensity, true);
      } else if (dtype == MLX_DTYPE_F32) {
        MLX_LAUNCH_K0(float, 4, kF32Faithful, kDensity, false);
      } else {
        MLX_LAUNCH_K0(float, 4, kF32Upcast, kDensity, false);
      }
#undef MLX_LAUNCH_K0
#undef MLX_LAUNCH_K0P
    } else {
      constexpr int U = 4;
      dim3 grid((unsigned)ceil_div(plane, (int64_t)kBlock * U), (unsigned)nz, (unsigned)ntc);
#define MLX_LAUNCH_K0G(TIN, MODE, FMA)                                                           \
  hipLaunchKernelGGL((k_eos_map<TIN, 1, U, MODE, 0, true, FMA>), grid, dim3(kBlock), 0, st,      \
                     (const TIN*)T, (const TIN*)S, pp, p_mode, eos, func, nz, plane, sT, sS, tb, \
                     aux, out)
      if (fma) {
        if (f64) MLX_LAUNCH_K0G(double, kF64, true);
        else if (dtype == MLX_DTYPE_F32) MLX_LAUNCH_K0G(float, kF32Faithful, true);
        else MLX_LAUNCH_K0G(float, kF32Upcast, true);
      } else if (f64) MLX_LAUNCH_K0G(double, kF64, false);
      else if (dtype == MLX_DTYPE_F32) MLX_LAUNCH_K0G(float, kF32Faithful, false);
      else MLX_LAUNCH_K0G(float, kF32Upcast, false);
#undef MLX_LAUNCH_K0G
    }
  }
  return hip_status(hipGetLastError(), "mlx_eos_map launch");
}

int mlx_eos_map(const void* T, const void* S, int dtype, const double* p, int p_mode, int eos,
                int func, int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS,
                int flags, double* out, void* stream) {
  if (func == MLX_FUNC_IBH) return fail(MLX_E_ENUM, "use mlx_inverse_barometer for MLX_FUNC_IBH");
  return eos_map_impl(T, S, dtype, p, p_mode, eos, func, 0.0, nt, nz, plane, sT, sS, flags, out,
                      stream);
}

int mlx_inverse_barometer(const void* T, const void* S, int dtype, const double* p, int p_mode,
                          int eos, double gravity, int64_t nt, int64_t nz, int64_t plane,
                          int64_t sT, int64_t sS, double* out, void* stream) {
  return eos_map_impl(T, S, dtype, p, p_mode, eos, MLX_FUNC_IBH, gravity, nt, nz, plane, sT, sS,
                      0, out, stream);
}

// ---------------------------------------------------------------------------- K1
size_t mlx_steric_global_workspace_bytes(int64_t nt, int64_t nz, int64_t plane) {
  if (nt <= 0 || nz <= 0 || plane <= 0) return 0;
  // the generic path has the smaller tile, hence the larger block count: size for it
  int64_t nblk, n;
  if (!mul_fits(ceil_div(plane, (int64_t)kBlock * kUGen), nz, &nblk) || !mul_fits(nt, nblk, &n))
    return 0;  // such a grid is rejected by the entry points (MLX_E_SHAPE)
  return (size_t)n * sizeof(double);
}

size_t mlx_steric_global_decomp_workspace_bytes(int64_t nt, int64_t nz, int64_t plane) {
  return 4 * mlx_steric_global_workspace_bytes(nt, nz, plane);
}

int mlx_steric_global(const void* T, const void* S, int dtype, const double* vol0, const double* p,
                      int p_mode, int eos, int64_t nt, int64_t nz, int64_t plane, int64_t sT,
                      int64_t sS, int flags, double* masso_out, void* workspace,
                      size_t workspace_bytes, void* stream) {
  return steric_global_impl(T, S, nullptr, nullptr, kVarSteric, dtype, vol0, p, p_mode, eos, nt,
                            nz, plane, sT, sS, flags, masso_out, workspace, workspace_bytes,
                            stream, "k_steric_global launch");
}

int mlx_steric_global_decomp(const void* T, const void* S, const void* T0, const void* S0,
                             int dtype, const double* vol0, const double* p, int p_mode, int eos,
                             int64_t nt, int64_t nz, int64_t plane, int64_t sT, int64_t sS,
                             int flags, double* out, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (sT == 0 || sS == 0)
    return fail(MLX_E_SHAPE, "mlx_steric_global_decomp streams both fields: time strides must be > 0");
  return steric_global_impl(T, S, T0, S0, kVarAll, dtype, vol0, p, p_mode, eos, nt, nz, plane, sT,
                            sS, flags, out, workspace, workspace_bytes, stream,
                            "k_steric_global (decomposition) launch");
}

// ---------------------------------------------------------------------------- K2
int mlx_fold_mask(const double* rho0, const double* vol0, int64_t n, double* rho0m_out,
                  void* stream) {
  if (!rho0 || !vol0 || !rho0m_out) return fail(MLX_E_NULL, "rho0, vol0, rho0m_out must not be NULL");
  if (n <= 0) return fail(MLX_E_SHAPE, "n must be > 0");
  if (!aligned(rho0, 8) || !aligned(vol0, 8) || !aligned(rho0m_out, 8))
    return fail(MLX_E_ALIGN, "operands not 8-byte aligned");
  const int64_t blocks = ceil_div(n, kBlock);
  hipLaunchKernelGGL(k_fold_mask, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(kBlock), 0,
                     (hipStream_t)stream, rho0, vol0, n, rho0m_out);
  return hip_status(hipGetLastError(), "k_fold_mask launch");
}

int mlx_steric_local(const void* T, const void* S, int dtype, const double* rho0m,
                     const double* vol0_surface, const double* dz, const double* z_i,
                     const double* deptho, const double* p, int p_mode, int eos,
                     double neg_inv_rhozero, int64_t nt, int64_t nz, int64_t plane, int64_t sT,
                     int64_t sS, int flags, double* delta_rho_out, double* eta_out, void* stream) {
  return steric_local_impl(T, S, nullptr, nullptr, kVarSteric, dtype, rho0m, vol0_surface, dz, z_i,
                           deptho, p, p_mode, eos, neg_inv_rhozero, nt, nz, plane, sT, sS, flags,
                           delta_rho_out, 0, eta_out, 0, stream);
}

int mlx_steric_local_decomp(const void* T, const void* S, const void* T0, const void* S0, int dtype,
                            const double* rho0m, const double* vol0_surface, const double* dz,
                            const double* z_i, const double* deptho, const double* p, int p_mode,
                            int eos, double neg_inv_rhozero, int64_t nt, int64_t nz, int64_t plane,
                            int64_t sT, int64_t sS, int flags, double* delta_rho_out,
                            int64_t delta_rho_variant_stride, double* eta_out,
                            int64_t eta_variant_stride, void* stream) {
  if (sT == 0 || sS == 0)
    return fail(MLX_E_SHAPE, "mlx_steric_local_decomp streams both fields: time strides must be > 0");
  return steric_local_impl(T, S, T0, S0, kVarAll, dtype, rho0m, vol0_surface, dz, z_i, deptho, p,
                           p_mode, eos, neg_inv_rhozero, nt, nz, plane, sT, sS, flags,
                           delta_rho_out, delta_rho_variant_stride, eta_out, eta_variant_stride,
                           stream);
}

// ---------------------------------------------------------------------------- sums
static int64_t nansum_blocks(int64_t n) {
  const int64_t b = ceil_div(n, (int64_t)kBlock * 8);
  return b < 1 ? 1 : (b > 8192 ? 8192 : b);
}

size_t mlx_nansum_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (size_t)nansum_blocks(n) * sizeof(double);
}

int mlx_nansum(const double* x, int64_t n, double* out, void* workspace, size_t workspace_bytes,
               void* stream) {
  if (!x || !out || !workspace) return fail(MLX_E_NULL, "x, out, workspace must not be NULL");
  if (n <= 0) return fail(MLX_E_SHAPE, "n must be > 0");
  if (!aligned(x, 8) || !aligned(out, 8)) return fail(MLX_E_ALIGN, "x/out not 8-byte aligned");
  if (!aligned(workspace, 8) || workspace_bytes < mlx_nansum_workspace_bytes(n))
    return fail(MLX_E_WORKSPACE, "workspace smaller than mlx_nansum_workspace_bytes()");
  const int64_t nb = nansum_blocks(n);
  hipStream_t st = (hipStream_t)stream;
  if (n % 2 == 0 && aligned(x, 16))
    hipLaunchKernelGGL(k_nansum_partial<true>, dim3((unsigned)nb), dim3(kBlock), 0, st, x, n,
                       (double*)workspace);
  else
    hipLaunchKernelGGL(k_nansum_partial<false>, dim3((unsigned)nb), dim3(kBlock), 0, st, x, n,
                       (double*)workspace);
  hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(kBlock), 0, st, (const double*)workspace, nb, out);
  return hip_status(hipGetLastError(), "mlx_nansum launch");
}

int mlx_masso(const double* rho, const double* vol, int64_t nt, int64_t n3, int64_t vol_t_stride,
              double* masso_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!rho || !vol || !masso_out || !workspace)
    return fail(MLX_E_NULL, "rho, vol, masso_out, workspace must not be NULL");
  if (nt <= 0 || n3 <= 0 || nt > 65535) return fail(MLX_E_SHAPE, "need 0 < nt <= 65535, n3 > 0");
  if (vol_t_stride != 0 && vol_t_stride != n3)
    return fail(MLX_E_SHAPE, "vol_t_stride must be 0 or n3");
  if (!aligned(rho, 8) || !aligned(vol, 8) || !aligned(masso_out, 8))
    return fail(MLX_E_ALIGN, "operands not 8-byte aligned");
  const int64_t nb = nansum_blocks(n3);
  if (!aligned(workspace, 8) || workspace_bytes < (size_t)(nt * nb) * sizeof(double) ||
      workspace_bytes < mlx_steric_global_workspace_bytes(nt, 1, n3))
    return fail(MLX_E_WORKSPACE, "workspace smaller than mlx_steric_global_workspace_bytes(nt,1,n3)");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_masso_partial, dim3((unsigned)nb, (unsigned)nt), dim3(kBlock), 0, st, rho,
                     vol, n3, vol_t_stride, (double*)workspace);
  hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)nt), dim3(kBlock), 0, st,
                     (const double*)workspace, nb, masso_out);
  return hip_status(hipGetLastError(), "mlx_masso launch");
}

int mlx_group_weighted_mean(const double* x, const double* w, int64_t ngroups, int64_t group_len,
                            int64_t n, double* out, void* stream) {
  if (!x || !w || !out) return fail(MLX_E_NULL, "x, w, out must not be NULL");
  if (ngroups <= 0 || group_len <= 0 || n <= 0 || ngroups > 65535)
    return fail(MLX_E_SHAPE, "need 0 < ngroups <= 65535, group_len > 0, n > 0");
  if (!aligned(x, 8) || !aligned(w, 8) || !aligned(out, 8))
    return fail(MLX_E_ALIGN, "operands not 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (n % 2 == 0 && aligned(x, 16) && aligned(out, 16)) {
    dim3 grid((unsigned)ceil_div(n, (int64_t)kBlock * 2), (unsigned)ngroups);
    hipLaunchKernelGGL(k_group_weighted_mean<2>, grid, dim3(kBlock), 0, st, x, w, group_len, n, out);
  } else {
    dim3 grid((unsigned)ceil_div(n, (int64_t)kBlock), (unsigned)ngroups);
    hipLaunchKernelGGL(k_group_weighted_mean<1>, grid, dim3(kBlock), 0, st, x, w, group_len, n, out);
  }
  return hip_status(hipGetLastError(), "k_group_weighted_mean launch");
}

int mlx_calc_dz(const double* z_i, const double* depth, int64_t nz, int64_t plane, double top,
                double bottom, int has_bottom, int fraction, double* dz_out, void* stream) {
  if (!z_i || !depth || !dz_out) return fail(MLX_E_NULL, "z_i, depth, dz_out must not be NULL");
  if (nz <= 0 || plane <= 0) return fail(MLX_E_SHAPE, "nz and plane must be > 0");
  if (!aligned(z_i, 8) || !aligned(depth, 8) || !aligned(dz_out, 8))
    return fail(MLX_E_ALIGN, "operands not 8-byte aligned");
  const int64_t gx = ceil_div(plane, kBlock);
  if (gx > 2147483647LL) return fail(MLX_E_SHAPE, "plane too large");
  hipLaunchKernelGGL(k_calc_dz, dim3((unsigned)gx), dim3(kBlock), 0, (hipStream_t)stream, z_i,
                     depth, nz, plane, top, bottom, has_bottom, fraction, dz_out);
  return hip_status(hipGetLastError(), "k_calc_dz launch");
}

int mlx_stream_probe_mix(const void* a, const void* b, int dtype, int64_t n, double* out,
                         int write_out, void* stream) {
  if (!a || !out) return fail(MLX_E_NULL, "a and out must not be NULL");
  if (dtype != MLX_DTYPE_F64 && dtype != MLX_DTYPE_F32)
    return fail(MLX_E_ENUM, "dtype must be MLX_DTYPE_F64 or MLX_DTYPE_F32");
  const int vec = (dtype == MLX_DTYPE_F64) ? 2 : 4;
  if (n <= 0 || n % vec) return fail(MLX_E_SHAPE, "n must be > 0 and a whole number of 16-byte packs");
  if (!aligned(a, 16) || (b && !aligned(b, 16)) || !aligned(out, write_out ? 16 : 8))
    return fail(MLX_E_ALIGN, "operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MLX_DTYPE_F64) {
    if (b) return write_out ? launch_probe_mix<double, 2, true, 1, true>(a, b, n, out, st)
                            : launch_probe_mix<double, 2, false, 2, true>(a, b, n, out, st);
    return write_out ? launch_probe_mix<double, 1, true, 1, true>(a, b, n, out, st)
                     : launch_probe_mix<double, 1, false, 2, true>(a, b, n, out, st);
  }
  if (b) return write_out ? launch_probe_mix<float, 2, true, 1, true, 2>(a, b, n, out, st)
                          : launch_probe_mix<float, 2, false, 8, true>(a, b, n, out, st);
  return write_out ? launch_probe_mix<float, 1, true, 1, true, 2>(a, b, n, out, st)
                   : launch_probe_mix<float, 1, false, 2, true>(a, b, n, out, st);
}

// the 2 x float64 in / 1 x float64 out mix under its round-2 name
int mlx_stream_probe(const double* a, const double* b, int64_t n, double* out, void* stream) {
  if (!a || !b || !out) return fail(MLX_E_NULL, "a, b, out must not be NULL");
  if (n <= 0 || n % 2) return fail(MLX_E_SHAPE, "n must be > 0 and even");
  return mlx_stream_probe_mix(a, b, MLX_DTYPE_F64, n, out, 1, stream);
}

int mlx_valu_probe(int64_t iters, double* out, int64_t* lane_instructions, void* stream) {
  if (!out || !lane_instructions) return fail(MLX_E_NULL, "out and lane_instructions must not be NULL");
  if (iters <= 0 || iters > (1 << 24)) return fail(MLX_E_SHAPE, "need 0 < iters <= 2^24");
  if (!aligned(out, 8)) return fail(MLX_E_ALIGN, "out not 8-byte aligned");
  *lane_instructions = (int64_t)kValuProbeBlocks * kBlock * 8 * iters;  // v_fma_f64, per lane
  hipLaunchKernelGGL(k_valu_probe, dim3(kValuProbeBlocks), dim3(kBlock), 0, (hipStream_t)stream,
                     out, (int)iters, 0.999999, 1.0e-6);
  return hip_status(hipGetLastError(), "k_valu_probe launch");
}

int mlx_synth_field(void* out, int dtype, int64_t nt, int64_t nz, int64_t ny, int64_t nx,
                    int64_t t0, int64_t NY, int64_t NX, int64_t y0, int64_t x0, uint64_t seed,
                    int field_id, double lo, double scale, const double* mask3d, void* stream) {
  if (!out) return fail(MLX_E_NULL, "out must not be NULL");
  if (nt <= 0 || nz <= 0 || ny <= 0 || nx <= 0) return fail(MLX_E_SHAPE, "dims must be > 0");
  if (y0 < 0 || x0 < 0 || t0 < 0 || NY <= 0 || NX <= 0 || ny > NY || nx > NX || y0 > NY - ny ||
      x0 > NX - nx)
    return fail(MLX_E_SHAPE, "tile does not fit the global grid");
  {  // the global cell counter ((t0+nt)*nz*NY*NX) must fit 63 bits
    int64_t a, b, c;
    if (t0 > INT64_MAX - nt || !mul_fits(t0 + nt, nz, &a) || !mul_fits(a, NY, &b) ||
        !mul_fits(b, NX, &c))
      return fail(MLX_E_SHAPE, "global index overflows");
  }
  if (field_id < 0 || field_id > 15) return fail(MLX_E_ENUM, "field_id must be 0..15");
  if (int rc = check_dtype(dtype, false)) return rc;
  if (!aligned(out, dtype == MLX_DTYPE_F64 ? 8 : 4) || (mask3d && !aligned(mask3d, 8)))
    return fail(MLX_E_ALIGN, "out/mask3d not element-aligned");
  const int64_t n = nt * nz * ny * nx;  // <= the global counter checked above
  const int64_t want = ceil_div(n, kBlock);
  dim3 grid((unsigned)(want < 16384 ? want : 16384));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MLX_DTYPE_F64)
    hipLaunchKernelGGL(k_synth<double>, grid, dim3(kBlock), 0, st, (double*)out, nt, nz, ny, nx,
                       t0, NY, NX, y0, x0, (unsigned long long)seed, field_id, lo, scale, mask3d);
  else
    hipLaunchKernelGGL(k_synth<float>, grid, dim3(kBlock), 0, st, (float*)out, nt, nz, ny, nx, t0,
                       NY, NX, y0, x0, (unsigned long long)seed, field_id, lo, scale, mask3d);
  return hip_status(hipGetLastError(), "k_synth launch");
}

}  // extern "C"
