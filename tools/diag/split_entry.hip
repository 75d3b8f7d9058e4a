// Diagnostic entry (not part of libumetrack_hip.so): one convolution through launch_conv_igemm (mode 0) or
// launch_conv_split (mode 1), and the host-side weight split.
#include <vector>

#include "ut_kernels.h"

namespace ut {      // tools/diag/conv_c64r.hip (lab: the one-wave-per-SIMD predecessor of conv_c64k.hip)
bool conv_c64r_applicable(const ConvLaunch& c);
hipError_t launch_conv_c64r(const ConvLaunch& c, hipStream_t s);
}

static float g_unscale = 1.f;
extern "C" int split_pack(const float* w, int cout_pad, int k_pad, uint16_t* out) {
  const float scale = ut::split_weight_scale(w, (size_t)cout_pad * k_pad);
  g_unscale = 1.f / scale;
  return (int)(ut::pack_split_weights(w, cout_pad, k_pad, scale, out) != (size_t)2 * cout_pad * k_pad);
}

static int* g_dbg = nullptr;            // stamp buffer of -DC64_STAMPS builds (8 x u64 per workgroup)
extern "C" int conv_stamps(unsigned long long* host) {
  if (!g_dbg) return 1;
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(host, g_dbg, 256 * 8 * 8, hipMemcpyDeviceToHost);
}

extern "C" int conv_diag2(const float* in, const float* w, const void* w_split, const float* bias, const float* res,
                          float* out, int n_img, int hw, int cin, int cout, int ksize, int stride, int relu, int mode) {
  ut::ConvLaunch c{};
  c.in = in; c.w = w; c.w_split = w_split; c.split_unscale = g_unscale; c.bias = bias; c.res = res; c.out = out;
  c.n_img = n_img; c.H = hw; c.W = hw; c.cin = cin;
  c.ksize = ksize; c.stride = stride; c.pad = ksize / 2;
  c.Ho = (hw + 2 * c.pad - ksize) / stride + 1; c.Wo = c.Ho;
  c.cout_store = cout; c.cout_pad = (cout + 127) / 128 * 128; c.k_total = ksize * ksize * cin; c.k_pad = c.k_total;
  c.cslice = cin % 32 == 0 ? 32 : cin; c.relu = relu; c.out_nchw = 0;
  c.num_cu = 256; c.device = 0;
  static unsigned* cnt = nullptr;
  if (!cnt) (void)hipMalloc((void**)&cnt, 4);
  (void)hipMemsetAsync(cnt, 0, 4, 0);
  c.tile_counter = cnt;
  if (!mode) c.w_split = nullptr;
  c.no_resident = mode == 3 ? 15 : mode == 4 ? 14 : 0;      // 3: the chunked kernels only, 4: conv_c64k where applicable, no conv_w4      // 1: conv_c64k where applicable, 2: conv_c64r (lab), 3: the chunked kernel
  if (mode == 2 && ut::conv_c64r_applicable(c)) return (int)ut::launch_conv_c64r(c, 0);
#ifdef C64_STAMPS
  if (!g_dbg) { (void)hipMalloc((void**)&g_dbg, 256 * 8 * 8); (void)hipMemset(g_dbg, 0, 256 * 8 * 8); }
  c.status = g_dbg;
#endif
  // mode 1: the split-fp16 kernel for the shape (conv_split.hip, or the split instantiation of the layer1 patch kernel)
  return (int)(mode && ut::conv_split_applicable(c) ? ut::launch_conv_split(c, 0) : ut::launch_conv_igemm(c, 0));
}
