// Checks the DPP / lane-swap reductions of csrc/common.h against plain sums on one wave.
// Build and run on the GPU box: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/dpp_reduce_probe.hip -o tools/probes/dpp_reduce_probe && ./tools/probes/dpp_reduce_probe
#include "../../self-supervised-wafermaps_amd/csrc/common.h"
#include <cstdio>
#include <vector>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float v = (float)(1 << (l % 20)) + l * 0.001f + 1.0f;
  out[0 * 64 + l] = group_sum<2>(v);
  out[1 * 64 + l] = group_sum<4>(v);
  out[2 * 64 + l] = group_sum<8>(v);
  out[3 * 64 + l] = group_sum<16>(v);
  out[4 * 64 + l] = group_sum<32>(v);
  out[5 * 64 + l] = group_sum<64>(v);
  out[6 * 64 + l] = group_max<64>(v);
  out[7 * 64 + l] = wm_xor16_sum(v);
  out[8 * 64 + l] = wm_xor32_sum(v);
  out[9 * 64 + l] = wm_dpp<0xB1>(v);
  out[10 * 64 + l] = wm_dpp<0x4E>(v);
  out[11 * 64 + l] = wm_dpp<0x141>(v);
  out[12 * 64 + l] = wm_dpp<0x140>(v);
  float a, b;
  wm_pair16(v, a, b);
  out[13 * 64 + l] = a;
  out[14 * 64 + l] = b;
  wm_pair32(v, a, b);
  out[15 * 64 + l] = a;
  out[16 * 64 + l] = b;
}
int main() {
  float* d;
  hipMalloc(&d, 17 * 64 * 4);
  k<<<1, 64>>>(d);
  std::vector<float> h(17 * 64);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  auto val = [](int l) { return (float)(1 << (l % 20)) + l * 0.001f + 1.0f; };
  const int widths[6] = {2, 4, 8, 16, 32, 64};
  for (int t = 0; t < 6; ++t) {
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      double s = 0;
      for (int j = 0; j < widths[t]; ++j) s += val((l / widths[t]) * widths[t] + j);
      if (fabs(h[t * 64 + l] - s) > 1e-3 * s) ++bad;
    }
    printf("group_sum<%d>: %d bad lanes\n", widths[t], bad);
  }
  const char* names[] = {"quad[1,0,3,2]", "quad[2,3,0,1]", "half_mirror", "mirror", "pair16.a", "pair16.b", "pair32.a", "pair32.b"};
  const int rows[] = {9, 10, 11, 12, 13, 14, 15, 16};
  for (int t = 0; t < 8; ++t) {
    printf("%s: source lanes:", names[t]);
    for (int l = 0; l < 64; ++l) {
      int src = -1;
      for (int j = 0; j < 64; ++j) if (h[rows[t] * 64 + l] == val(j)) src = j;
      printf(" %d", src);
    }
    printf("\n");
  }
  return 0;
}
