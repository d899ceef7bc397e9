// Shared by the translation units of libmomlevel_hip.so; nothing here is part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

namespace mlx {
namespace detail {
// record the text mlx_last_error() returns on this thread and hand back `code`
__attribute__((visibility("hidden"))) int fail(int code, const char* msg);
// 0 for hipSuccess; otherwise records "<what>: <hip error string>" and returns the hipError_t
__attribute__((visibility("hidden"))) int hip_status(hipError_t e, const char* what);
}  // namespace detail
}  // namespace mlx
