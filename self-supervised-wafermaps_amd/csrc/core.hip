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

// ---- the small device-side pieces that would otherwise be framework (ATen) launches inside the captured training
// step: clearing the flat gradient arena (optimizer.zero_grad) and scalar means of short vectors (the mean of the
// NT-Xent row losses; the mean of sqrt(scale * column variance) of lightly's std_of_l2_normalized monitor).
namespace {
// out[0] = mean_i f(x_i), f = identity (sqrt_of = 0) or sqrt(scale * x_i).  ONE block, a fixed summation order
// (strided per-thread sums, then a fixed tree): bit-reproducible.
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ x, long long n, float scale, int sqrt_of,
                                                    float* __restrict__ out) {
  __shared__ float red[1024];
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) s += sqrt_of ? sqrtf(scale * x[i]) : x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (float)n;
}
}  // namespace

namespace {
// y = x * *scale (bf16 -> bf16 through f32): the upstream gradient of a scalar loss applied to a saved gradient tensor
__global__ __launch_bounds__(256) void scale_bf16_kernel(const uint16_t* __restrict__ x, long long n8, const float* __restrict__ scale,
                                                        uint16_t* __restrict__ y) {
  const float s = *scale;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4*>(x)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      o[q] = pack_bf2(bf2f((uint16_t)(w[q] & 0xffff)) * s, bf2f((uint16_t)(w[q] >> 16)) * s);
    reinterpret_cast<uint4*>(y)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
}  // namespace

extern "C" int wm_scale_bf16(const void* x, long long n, const float* scale_dev, void* y, void* stream) {
  WM_REQUIRE(x && y && scale_dev && n > 0, WM_EINVAL);
  WM_REQUIRE(n % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, WM_EALIGN);
  const long long n8 = n / 8;
  const long long blocks = (n8 + 255) / 256;
  scale_bf16_kernel<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(x), n8, scale_dev, static_cast<uint16_t*>(y));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_fill_zero(void* p, size_t bytes, void* stream) {
  WM_REQUIRE(p && bytes > 0, WM_EINVAL);
  WM_REQUIRE(bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0, WM_EALIGN);
  const hipError_t e = wm_zero_async(p, bytes, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? WM_OK : (int)e;
}

extern "C" int wm_mean_f32(const float* x, long long n, float scale, int sqrt_of, float* out, void* stream) {
  WM_REQUIRE(x && out && n > 0, WM_EINVAL);
  mean_kernel<<<1, 1024, 0, static_cast<hipStream_t>(stream)>>>(x, n, scale, sqrt_of, out);
  WM_LAUNCH_CHECK();
  return WM_OK;
}
