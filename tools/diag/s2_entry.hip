// Diagnostic entry (not part of libumetrack_hip.so): layer2's stride-2 entry (conv_c32s2.hip) on its own, and the host-side weight split.
#include <vector>

#include "ut_kernels.h"

static float g_unscale[2] = {1.f, 1.f};
extern "C" int s2_pack(int which, const float* w, int cout_pad, int k_pad, uint16_t* out) {
  const float scale = ut::split_weight_scale(w, (size_t)cout_pad * k_pad);
  g_unscale[which] = 1.f / scale;
  return (int)(ut::pack_split_weights(w, cout_pad, k_pad, scale, out) != (size_t)2 * cout_pad * k_pad);
}

static int* g_dbg = nullptr;            // stamp buffer of patched builds (8 x u64 per workgroup)
extern "C" int s2_stamps(unsigned long long* host) {
  if (!g_dbg) return 1;
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(host, g_dbg, 256 * 8 * 8, hipMemcpyDeviceToHost);
}

extern "C" int s2_diag(const float* in, const void* w1_split, const void* wd_split, const float* bias1, const float* bias_d, float* out1,
                       float* out2, int n_img, int H, int W, const unsigned* in_max, unsigned* out_max, int stamps) {
  ut::Stride2Launch c{};
  c.in = in; c.out1 = out1; c.out2 = out2; c.w1_split = w1_split; c.wd_split = wd_split;
  c.unscale1 = g_unscale[0]; c.unscale_d = g_unscale[1]; c.bias1 = bias1; c.bias_d = bias_d;
  c.in_max = in_max; c.out1_max = out_max; c.n_img = n_img; c.H = H; c.W = W; c.device = 0; c.num_cu = 256;
  if (stamps) {
    if (!g_dbg) { (void)hipMalloc((void**)&g_dbg, 256 * 8 * 8); (void)hipMemset(g_dbg, 0, 256 * 8 * 8); }
    c.status = g_dbg;
  }
  (void)hipMemsetAsync(out_max, 0, 4, 0);
  if (!ut::conv_c32s2_applicable(c)) return -1;
  return (int)ut::launch_conv_c32s2(c, 0);
}
