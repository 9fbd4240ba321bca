// Library identity + error strings for the wafer_hip C ABI.
#include "common.h"

extern "C" int wm_version(void) { return WM_ABI_VERSION; }

extern "C" const char* wm_error_string(int code) {
  switch (code) {
    case WM_OK: return "ok";
    case WM_EINVAL: return "invalid argument (null pointer or non-positive size)";
    case WM_EUNSUPPORTED: return "unsupported shape/dtype combination";
    case WM_EWORKSPACE: return "workspace too small";
    case WM_EALIGN: return "pointer or pitch alignment requirement not met";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown wafer_hip error";
}

// ---- debugging probe: max |x| of a tensor into one float slot, NaN-propagating, no allocation (so it can be
// enqueued between the launches of a captured hipGraph without changing the caller's memory layout).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void absmax_probe(const T* __restrict__ x, long long n, uint32_t* __restrict__ slot) {
  uint32_t m = 0;  // bit pattern of a non-negative float: unsigned order == float order, NaN (0x7FC00000) on top
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v;
    if constexpr (sizeof(T) == 2) v = bf2f(x[i]); else v = x[i];
    const uint32_t b = (v != v) ? 0x7FC00000u : (__builtin_bit_cast(uint32_t, v) & 0x7FFFFFFFu);
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)m, o, 64);
    m = other > m ? other : m;
  }
  if ((threadIdx.x & 63) == 0 && m != 0) atomicMax(slot, m);
}
}  // namespace

extern "C" int wm_debug_absmax(const void* x, int dtype, long long n, float* slot, void* stream) {
  WM_REQUIRE(x && slot && n > 0, WM_EINVAL);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  if (dtype == WM_F32)
    absmax_probe<float><<<grid, 256, 0, st>>>((const float*)x, n, reinterpret_cast<uint32_t*>(slot));
  else if (dtype == WM_BF16)
    absmax_probe<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)x, n, reinterpret_cast<uint32_t*>(slot));
  else
    return WM_EUNSUPPORTED;
  WM_LAUNCH_CHECK();
  return WM_OK;
}
