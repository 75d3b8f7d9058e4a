// Diagnostic entry (not part of libumetrack_hip.so): one stride-1 3x3 convolution through launch_conv_igemm.

#include "ut_kernels.h"
#ifdef DIAG_WS
namespace ut { hipError_t launch_conv_ws(const ConvLaunch& c, hipStream_t s); }
#endif
extern "C" int conv_diag(const float* in, const float* w, const float* bias, const float* res, float* out, int n_img,
                         int hw, int cin, int cout, int k_total, long long* stamps) {
  ut::ConvLaunch c{};
  c.in = in; c.w = w; c.bias = bias; c.res = res; c.out = out;
  c.n_img = n_img; c.H = hw; c.W = hw; c.cin = cin; c.Ho = hw; c.Wo = hw;
  c.cout_store = cout; c.cout_pad = (cout + 127) / 128 * 128; c.k_total = k_total; c.k_pad = k_total;
  c.cslice = cin % 32 == 0 ? 32 : cin; c.ksize = 3; c.stride = 1; c.pad = 1; c.relu = 1; c.out_nchw = 0;
  c.num_cu = 256; c.device = 0;
  (void)stamps;
  static unsigned* cnt = nullptr;
  if (!cnt) (void)hipMalloc((void**)&cnt, 4);
  (void)hipMemsetAsync(cnt, 0, 4, 0);
  c.tile_counter = cnt;
#ifdef DIAG_WS
  return (int)ut::launch_conv_ws(c, 0);
#else
  return (int)ut::launch_conv_igemm(c, 0);
#endif
}
