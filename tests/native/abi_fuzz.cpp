// abi_fuzz.cpp -- argument fuzzer for the HOST side of libmomlevel_hip.so (SURVEY.md section 5,
// "race detection / sanitizers": the reference has only CodeQL; this build runs the C ABI's host
// code -- argument checks, launch-geometry arithmetic, workspace sizing, dispatch -- under
// AddressSanitizer + UndefinedBehaviorSanitizer).  CPU CONTAINER ONLY: the pointers are fakes.
// In a GPU-less process every call that survives the argument checks ends in
// hipErrorNoDevice (100) from the launch; with a device visible the program refuses to run.
//
//   scripts/sanitize_host.py   builds the sanitized library and this driver, then runs it.
//
// Contract checked for every call: the status is an MLX_E_* code (-1..-5) or a hipError_t (> 0),
// never 0 for a launching entry point here, a non-zero status leaves a message in
// mlx_last_error(), and nothing trips ASan/UBSan (both fatal).
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/momlevel_hip.h"

static uint64_t rng_state = 0x1234ABCDULL;
static uint64_t rnd() {  // splitmix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
template <typename T, size_t N>
static T pick(const T (&a)[N]) { return a[rnd() % N]; }

// "sane" mode: most draws are valid (small extents, aligned pointers, known enums) so that the
// calls get PAST the argument checks into plan / workspace / grid arithmetic; each picker still
// corrupts its value now and then.  The other half of the iterations draws wild values throughout.
static bool sane = false;
static bool corrupt() { return !sane || rnd() % 12 == 0; }

static int64_t dim() {
  if (!corrupt()) {
    static const int64_t ok[] = {1, 2, 3, 4, 8, 12, 32, 64, 75, 120, 1000, 4096, 65535, 1555200};
    return pick(ok);
  }
  static const int64_t v[] = {-7, -1, 0, 1, 1, 2, 3, 4, 5, 7, 8, 12, 31, 32, 33, 64, 75, 120, 1000,
                              2048, 65535, 65536, 1555200, 2147483647LL, 2147483648LL,
                              (int64_t)1 << 40, ((int64_t)1 << 40) + 1, (int64_t)1 << 62,
                              INT64_MAX, INT64_MIN};
  return pick(v);
}
static int64_t stride(int64_t n3) {
  if (!corrupt()) return (rnd() % 4 == 0) ? 0 : n3;
  static const int64_t v[] = {-1, 0, 0, 1, 2, 3, 4, 8, 1000, (int64_t)1 << 33};
  return (rnd() % 3 == 0) ? n3 : pick(v);
}
static void* ptr() {
  if (!corrupt()) return (void*)(uintptr_t)(0x10000 + 16 * (rnd() % 4096));
  static const uintptr_t v[] = {0, 0x10000, 0x10000, 0x10000, 0x10008, 0x10004, 0x10002, 0x10001,
                                0x7f0000000000ULL, 0x7f0000000010ULL, ~(uintptr_t)0 - 15};
  return (void*)pick(v);
}
static int small_enum() { return corrupt() ? (int)(rnd() % 9) - 2 : (int)(rnd() % 2); }
static int flagbits() {
  if (!corrupt()) { static const int ok[] = {0, 1, 2, 3, 0x800, 0x803}; return pick(ok); }
  static const int v[] = {0, 0, 1, 2, 3, 4, 0x100, 0x800, 0xFF00, 0xFF03, -1, 1 << 16, 0x7FFFFFFF};
  return pick(v);
}
static size_t nbytes(size_t need) {
  if (!corrupt()) return need;
  switch (rnd() % 5) {
    case 0: return 0;
    case 1: return need ? need - 1 : 0;
    case 2: return need;
    case 3: return need + 8;
    default: return ~(size_t)0;
  }
}

static long n_calls = 0, n_arg_errors = 0, n_launch_attempts = 0;
static void check(int rc, const char* what) {
  ++n_calls;
  char buf[512];
  if (rc == 0) {
    fprintf(stderr, "FAIL: %s returned 0 with fake pointers and no device\n", what);
    exit(1);
  }
  if (rc < 0) {
    if (rc < MLX_E_ALIGN) {
      fprintf(stderr, "FAIL: %s returned unknown argument-error code %d\n", what, rc);
      exit(1);
    }
    ++n_arg_errors;
  } else {
    ++n_launch_attempts;
  }
  if (mlx_last_error(buf, sizeof buf) <= 0 || buf[0] == 0) {
    fprintf(stderr, "FAIL: %s -> %d left no error text\n", what, rc);
    exit(1);
  }
}

int main(int argc, char** argv) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
    printf("SKIP: a HIP device is visible; the fuzzer passes fake pointers and only runs in a "
           "GPU-less process\n");
    return 77;
  }
  const long iters = argc > 1 ? atol(argv[1]) : 20000;
  if (mlx_version() != MLX_ABI_VERSION) return 2;
  {  // a NULL pressure (linear EOS) is a placeholder that only MLX_P_SCALAR never dereferences:
     // every array mode must be refused before anything is launched
    void* fake = (void*)(uintptr_t)(1 << 20);
    for (int pm = MLX_P_ZPROF; pm <= MLX_P_FULL4D; ++pm) {
      const int rc1 = mlx_steric_global(fake, fake, MLX_DTYPE_F64, (const double*)fake, nullptr, pm,
                                        MLX_EOS_LINEAR, 2, 3, 32, 96, 96, 0, (double*)fake, fake,
                                        1 << 20, nullptr);
      const int rc2 = mlx_steric_local(fake, fake, MLX_DTYPE_F64, (const double*)fake,
                                       (const double*)fake, (const double*)fake, nullptr, nullptr,
                                       nullptr, pm, MLX_EOS_LINEAR, -1.0 / 1035.0, 2, 3, 32, 96, 96,
                                       0, nullptr, (double*)fake, nullptr);
      if (rc1 != MLX_E_NULL || rc2 != MLX_E_NULL) {
        fprintf(stderr, "FAIL: NULL p with p_mode %d -> %d / %d, expected MLX_E_NULL\n", pm, rc1, rc2);
        return 1;
      }
    }
  }
  {  // mlx_host_copy works on REAL host memory: heap buffers with ASan's red zones either side,
     // alignments of source and destination, lengths around the 128-byte inner block, and sizes
     // that are split over the thread team (slices of >= 1 MiB)
    const size_t cap = (size_t)9 << 20;
    unsigned char* src = (unsigned char*)malloc(cap + 64);
    unsigned char* dst = (unsigned char*)malloc(cap + 64);
    for (size_t i = 0; i < cap + 64; ++i) src[i] = (unsigned char)(i * 131 + (i >> 12) + 7);
    static const size_t lens[] = {0, 1, 31, 32, 33, 127, 128, 129, 4095, 4096, 4097, 4223, 8192,
                                  65535, 69999, ((size_t)1 << 20) + 1, ((size_t)3 << 20) - 5,
                                  (size_t)9 << 20};
    static const int teams[] = {1, 2, 3, 8, 64};
    for (int streaming = 0; streaming < 2; ++streaming)
      for (int threads : teams)
        for (size_t so = 0; so < 40; so += 13)
          for (size_t dof = 0; dof < 40; dof += 11)
            for (size_t n : lens) {
              if (n < 4096 && threads > 2) continue;  // (one slice anyway)
              dst[dof + n] = 0xEE;
              if (dof) dst[dof - 1] = 0xEE;
              if (n) memset(dst + dof, 0, n < 4096 ? n : 4096);
              if (mlx_host_copy(dst + dof, src + so, n, threads, streaming) != 0 ||
                  memcmp(dst + dof, src + so, n) != 0 || dst[dof + n] != 0xEE ||
                  (dof && dst[dof - 1] != 0xEE)) {
                fprintf(stderr, "FAIL: mlx_host_copy(streaming=%d, threads=%d, src+%zu, dst+%zu, %zu)\n",
                        streaming, threads, so, dof, n);
                return 1;
              }
            }
    if (mlx_host_copy(nullptr, src, 8, 1, 1) != MLX_E_NULL ||
        mlx_host_copy(dst, nullptr, 8, 1, 1) != MLX_E_NULL ||
        mlx_host_copy(src + 8, src, 64, 1, 1) != MLX_E_SHAPE ||
        mlx_host_copy(src, src + 8, 64, 2, 0) != MLX_E_SHAPE ||
        mlx_host_copy(dst, src, 64, 0, 0) != MLX_E_SHAPE ||
        mlx_host_copy(dst, src, 64, 65, 0) != MLX_E_SHAPE ||
        mlx_host_copy(dst, (void*)(~(uintptr_t)0 - 15), 64, 1, 0) != MLX_E_SHAPE ||
        mlx_host_copy(nullptr, nullptr, 0, 1, 1) != 0) {
      fprintf(stderr, "FAIL: mlx_host_copy argument checks\n");
      return 1;
    }
    // mlx_host_copy_masked on the same buffers: element sizes, offsets, lengths around slice and
    // vector-width edges, team sizes; guard elements either side
    unsigned char* mask = (unsigned char*)malloc(cap + 64);
    for (size_t i = 0; i < cap + 64; ++i) mask[i] = (unsigned char)(((i * 2654435761u) >> 7) % 3 == 0);
    for (int elem = 4; elem <= 8; elem += 4)
      for (int threads : teams)
        for (size_t off = 0; off < 3; ++off)
          for (size_t n : lens) {
            const size_t cnt = n / (size_t)elem > 8 ? n / (size_t)elem - 3 * off : n % 7;
            if (cnt < 4096 && threads > 2) continue;
            unsigned char* d0 = dst + (1 + off) * (size_t)elem;
            const unsigned char* s0 = src + off * (size_t)elem;
            memset(d0 - elem, 0xEE, (size_t)elem);
            memset(d0 + cnt * (size_t)elem, 0xEE, (size_t)elem);
            if (mlx_host_copy_masked(d0, s0, mask + off, cnt, elem, threads) != 0) {
              fprintf(stderr, "FAIL: mlx_host_copy_masked rc (elem %d threads %d n %zu)\n", elem, threads, cnt);
              return 1;
            }
            for (size_t i = 0; i < cnt; ++i) {
              bool ok;
              if (elem == 4) {
                uint32_t got, want; memcpy(&got, d0 + 4 * i, 4); memcpy(&want, s0 + 4 * i, 4);
                ok = got == (mask[off + i] ? 0x7FC00000u : want);
              } else {
                uint64_t got, want; memcpy(&got, d0 + 8 * i, 8); memcpy(&want, s0 + 8 * i, 8);
                ok = got == (mask[off + i] ? 0x7FF8000000000000ull : want);
              }
              if (!ok) {
                fprintf(stderr, "FAIL: mlx_host_copy_masked value (elem %d threads %d n %zu i %zu)\n", elem, threads, cnt, i);
                return 1;
              }
            }
            if (d0[-1] != 0xEE || d0[cnt * (size_t)elem] != 0xEE) {
              fprintf(stderr, "FAIL: mlx_host_copy_masked wrote outside its range\n");
              return 1;
            }
          }
    if (mlx_host_copy_masked(nullptr, src, mask, 8, 8, 1) != MLX_E_NULL ||
        mlx_host_copy_masked(dst, src, nullptr, 8, 8, 1) != MLX_E_NULL ||
        mlx_host_copy_masked(dst, src, mask, 8, 3, 1) != MLX_E_ENUM ||
        mlx_host_copy_masked(dst, src, mask, 8, 8, 65) != MLX_E_SHAPE ||
        mlx_host_copy_masked(src + 8, src, mask, 8, 8, 1) != MLX_E_SHAPE ||
        mlx_host_copy_masked(dst, src, (unsigned char*)dst + 8, 8, 8, 1) != MLX_E_SHAPE ||
        mlx_host_copy_masked(dst, src, mask, (size_t)-1 / 4, 8, 1) != MLX_E_SHAPE ||
        mlx_host_copy_masked(nullptr, nullptr, nullptr, 0, 8, 1) != 0) {
      fprintf(stderr, "FAIL: mlx_host_copy_masked argument checks\n");
      return 1;
    }
    free(mask);
    free(src);
    free(dst);
  }
  char tiny[4];
  mlx_last_error(tiny, sizeof tiny);  // truncation path
  mlx_last_error(nullptr, 0);
  for (long it = 0; it < iters; ++it) {
    sane = (it % 2 == 1);
    const int64_t nt = dim(), nz = dim(), plane = dim();
    // the products may overflow int64 for the extreme draws: computed unsigned, only as a hint
    const int64_t n3 = (int64_t)((uint64_t)nz * (uint64_t)plane);
    const int dtype = small_enum(), p_mode = small_enum(), eos = small_enum(), func = small_enum();
    switch (rnd() % 15) {
      case 0:
        check(mlx_eos_map(ptr(), ptr(), dtype, (const double*)ptr(), p_mode, eos, func, nt, nz,
                          plane, stride(n3), stride(n3), flagbits(), (double*)ptr(), nullptr),
              "mlx_eos_map");
        break;
      case 1:
        check(mlx_inverse_barometer(ptr(), ptr(), dtype, (const double*)ptr(), p_mode, eos, 9.8,
                                    nt, nz, plane, stride(n3), stride(n3), (double*)ptr(), nullptr),
              "mlx_inverse_barometer");
        break;
      case 2: {
        const size_t need = mlx_steric_global_workspace_bytes(nt, nz, plane);
        check(mlx_steric_global(ptr(), ptr(), dtype, (const double*)ptr(), (const double*)ptr(),
                                p_mode, eos, nt, nz, plane, stride(n3), stride(n3), flagbits(),
                                (double*)ptr(), ptr(), nbytes(need), nullptr),
              "mlx_steric_global");
        break;
      }
      case 3: {
        const size_t need = mlx_steric_global_decomp_workspace_bytes(nt, nz, plane);
        check(mlx_steric_global_decomp(ptr(), ptr(), ptr(), ptr(), dtype, (const double*)ptr(),
                                       (const double*)ptr(), p_mode, eos, nt, nz, plane,
                                       stride(n3), stride(n3), flagbits(), (double*)ptr(), ptr(),
                                       nbytes(need), nullptr),
              "mlx_steric_global_decomp");
        break;
      }
      case 4:
        check(mlx_fold_mask((const double*)ptr(), (const double*)ptr(), dim(), (double*)ptr(),
                            nullptr),
              "mlx_fold_mask");
        break;
      case 5:
        check(mlx_steric_local(ptr(), ptr(), dtype, (const double*)ptr(), (const double*)ptr(),
                               (const double*)ptr(), (const double*)ptr(), (const double*)ptr(),
                               (const double*)ptr(), p_mode, eos, -1.0 / 1035.0, nt, nz, plane,
                               stride(n3), stride(n3), flagbits(), (double*)ptr(), (double*)ptr(),
                               nullptr),
              "mlx_steric_local");
        break;
      case 11:
        check(mlx_steric_local_decomp(ptr(), ptr(), ptr(), ptr(), dtype, (const double*)ptr(),
                                      (const double*)ptr(), (const double*)ptr(),
                                      (const double*)ptr(), (const double*)ptr(),
                                      (const double*)ptr(), p_mode, eos, -1.0 / 1035.0, nt, nz,
                                      plane, stride(n3), stride(n3), flagbits(), (double*)ptr(),
                                      (rnd() % 2) ? (int64_t)((uint64_t)nt * (uint64_t)n3) : dim(),
                                      (double*)ptr(),
                                      (rnd() % 2) ? (int64_t)((uint64_t)nt * (uint64_t)plane) : dim(),
                                      nullptr),
              "mlx_steric_local_decomp");
        break;
      case 13:
        check(mlx_stratification(ptr(), ptr(), dtype, (const double*)ptr(), stride(n3), stride(plane),
                                 corrupt() ? (int64_t)(rnd() % 5) - 1 : (int64_t)(rnd() % 2), eos,
                                 small_enum(), (const double*)ptr(), (int)(rnd() % 2),
                                 corrupt() ? 0.0 : 20.0, -9.8, nt, nz, plane, (double*)ptr(), nullptr),
              "mlx_stratification");
        break;
      case 14:
        if (rnd() % 2)
          check(mlx_adjust_negative_n2((const double*)ptr(), nt, nz, plane,
                                       corrupt() ? dim() : (int64_t)(rnd() % 2),
                                       (const double*)ptr(), (double*)ptr(), (double*)ptr(), nullptr),
                "mlx_adjust_negative_n2");
        else
          check(mlx_wave_speed_where_time0((const double*)ptr(), (const double*)ptr(), nt, nz, plane,
                                           (double*)ptr(), nullptr),
                "mlx_wave_speed_where_time0");
        break;
      case 12: {
        // a weak operand is a HOST pointer the call reads: a real double (or NULL), never a fake
        static const double host_scalar = 101325.0;
        int kinds[3];
        const void* ops[3];
        for (int k = 0; k < 3; ++k) {
          kinds[k] = corrupt() ? (int)(rnd() % 9) - 3 : (int)(rnd() % 3);
          ops[k] = (kinds[k] == MLX_KIND_WEAK) ? ((rnd() % 8) ? (const void*)&host_scalar : nullptr)
                                               : (const void*)ptr();
        }
        int out_kind = -1;
        check(mlx_eos_map_promote(ops[0], kinds[0], stride(1), ops[1], kinds[1], stride(1), ops[2],
                                  kinds[2], stride(1), eos, corrupt() ? (int)(rnd() % 12) - 3 : (int)(rnd() % 6),
                                  9.8, dim(), ptr(), (rnd() % 8) ? &out_kind : nullptr,
                                  nullptr),
              "mlx_eos_map_promote");
        break;
      }
      case 6: {
        const int64_t n = dim();
        check(mlx_nansum((const double*)ptr(), n, (double*)ptr(), ptr(),
                         nbytes(mlx_nansum_workspace_bytes(n)), nullptr),
              "mlx_nansum");
        break;
      }
      case 7:
        check(mlx_masso((const double*)ptr(), (const double*)ptr(), nt, n3, stride(n3),
                        (double*)ptr(), ptr(), nbytes(mlx_steric_global_workspace_bytes(nt, 1, n3)),
                        nullptr),
              "mlx_masso");
        break;
      case 8:
        check(mlx_group_weighted_mean((const double*)ptr(), (const double*)ptr(), dim(), dim(),
                                      dim(), (double*)ptr(), nullptr),
              "mlx_group_weighted_mean");
        break;
      case 9:
        check(mlx_calc_dz((const double*)ptr(), (const double*)ptr(), nz, plane, 0.0, 100.0,
                          (int)(rnd() % 2), (int)(rnd() % 2), (double*)ptr(), nullptr),
              "mlx_calc_dz");
        break;
      default:
        if (rnd() % 3 == 0)
          check(mlx_stream_probe((const double*)ptr(), (const double*)ptr(), dim(), (double*)ptr(),
                                 nullptr),
                "mlx_stream_probe");
        else if (rnd() % 4 == 0) {
            int64_t n = 0;
            check(mlx_valu_probe((int64_t)dim(), (double*)ptr(), corrupt() ? nullptr : &n, nullptr),
                  "mlx_valu_probe");
        }
        else if (rnd() % 2)
          check(mlx_stream_probe_mix(ptr(), ptr(), dtype, dim(), (double*)ptr(), (int)(rnd() % 2),
                                     nullptr),
                "mlx_stream_probe_mix");
        else
          check(mlx_synth_field(ptr(), dtype, nt, nz, dim(), dim(), dim(), dim(), dim(), dim(),
                                dim(), rnd(), small_enum(), -2.0, 34.0, (const double*)ptr(),
                                nullptr),
                "mlx_synth_field");
        break;
    }
  }
  printf("abi_fuzz OK: %ld calls, %ld argument errors, %ld reached the launch (hipErrorNoDevice)\n",
         n_calls, n_arg_errors, n_launch_attempts);
  return (n_arg_errors > 0 && n_launch_attempts > 0) ? 0 : 3;
}
