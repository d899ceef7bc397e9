// c_abi_demo.cpp -- the C ABI of libmomlevel_hip.so used from plain C++/HIP: no Python, no torch.
//
//   hipcc --offload-arch=gfx950 -O2 examples/c_abi_demo.cpp -Iinclude -Lmomlevel_amd \
//         -lmomlevel_hip -Wl,-rpath,'$ORIGIN/../momlevel_amd' -o examples/c_abi_demo
//   ./examples/c_abi_demo
//
// Builds a small synthetic (time, z_l, yh, xh) record on the device with mlx_synth_field, runs the
// fused global steric pass (mlx_steric_global) and the volume sum (mlx_nansum), and prints
// masso(t), volo and the global steric height h_ref * ln(rhoga0 * volo / masso(t)) for a unit area
// sum.  tests/test_gpu_c_abi.py runs it and compares the printed numbers with the Python path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "momlevel_hip.h"

#define HIP_OK(x)                                                         \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      fprintf(stderr, "HIP error: %s\n", hipGetErrorString(e_));          \
      return 2;                                                           \
    }                                                                     \
  } while (0)
#define MLX_OK(x)                                                         \
  do {                                                                    \
    int rc_ = (x);                                                        \
    if (rc_ != 0) {                                                       \
      char buf[512];                                                      \
      mlx_last_error(buf, sizeof buf);                                    \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, buf);                    \
      return 3;                                                           \
    }                                                                     \
  } while (0)

int main() {
  const int64_t nt = 6, nz = 10, ny = 32, nx = 48, plane = ny * nx, n3 = nz * plane;
  if (mlx_version() != MLX_ABI_VERSION) return 1;

  // pressure profile p = z_l*1e4 + patm on a 10-level grid; a volcello with ~25 % dry cells
  std::vector<double> pz(nz), vol(n3);
  for (int64_t k = 0; k < nz; ++k) pz[k] = (5.0 + 50.0 * k) * 1.0e4 + 101325.0;
  for (int64_t i = 0; i < n3; ++i) vol[i] = ((i / 7) % 4 == 0) ? NAN : 1.0e9 + 1.0e6 * (i % 1000);

  double *T, *S, *dvol, *dpz, *masso, *volo, *ws;
  HIP_OK(hipMalloc(&T, nt * n3 * 8));
  HIP_OK(hipMalloc(&S, nt * n3 * 8));
  HIP_OK(hipMalloc(&dvol, n3 * 8));
  HIP_OK(hipMalloc(&dpz, nz * 8));
  HIP_OK(hipMalloc(&masso, nt * 8));
  HIP_OK(hipMalloc(&volo, 8));
  HIP_OK(hipMemcpy(dvol, vol.data(), n3 * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dpz, pz.data(), nz * 8, hipMemcpyHostToDevice));
  size_t ws_bytes = mlx_steric_global_workspace_bytes(nt, nz, plane);
  const size_t ns_bytes = mlx_nansum_workspace_bytes(n3);
  if (ns_bytes > ws_bytes) ws_bytes = ns_bytes;
  HIP_OK(hipMalloc(&ws, ws_bytes));

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  // theta in [-2, 32), S in [30, 40), NaN where volcello is NaN
  MLX_OK(mlx_synth_field(T, MLX_DTYPE_F64, nt, nz, ny, nx, 0, ny, nx, 0, 0, 20251114ULL, 1, -2.0, 34.0,
                         dvol, stream));
  MLX_OK(mlx_synth_field(S, MLX_DTYPE_F64, nt, nz, ny, nx, 0, ny, nx, 0, 0, 20251114ULL, 2, 30.0, 10.0,
                         dvol, stream));
  MLX_OK(mlx_steric_global(T, S, MLX_DTYPE_F64, dvol, dpz, MLX_P_ZPROF, MLX_EOS_WRIGHT, nt, nz, plane,
                           n3, n3, MLX_FLAG_SKIP_DRY, masso, ws, ws_bytes, stream));
  MLX_OK(mlx_nansum(dvol, n3, volo, ws, ws_bytes, stream));
  HIP_OK(hipStreamSynchronize(stream));

  std::vector<double> m(nt);
  double v;
  HIP_OK(hipMemcpy(m.data(), masso, nt * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&v, volo, 8, hipMemcpyDeviceToHost));
  const double rhoga0 = m[0] / v;  // reference state = time index 0
  printf("volo %.17g\n", v);
  for (int64_t t = 0; t < nt; ++t)
    printf("t=%lld masso %.17g expansion %.17g\n", (long long)t, m[t], std::log(rhoga0 / (m[t] / v)));

  // an argument error is reported, not thrown
  const int rc = mlx_steric_global(T, S, 99, dvol, dpz, MLX_P_ZPROF, MLX_EOS_WRIGHT, nt, nz, plane, n3, n3,
                                   0, masso, ws, ws_bytes, stream);
  char buf[256];
  mlx_last_error(buf, sizeof buf);
  printf("bad dtype -> %d (%s)\n", rc, buf);
  return (rc == MLX_E_ENUM) ? 0 : 4;
}
