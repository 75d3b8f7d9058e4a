// Diagnostic entry (not part of libumetrack_hip.so): one fused layer1 BasicBlock (conv_block32.hip) on synthetic data.
#include <math.h>
#include <vector>

#include "ut_kernels.h"

static float g_unscale[2] = {1.f, 1.f};
static float g_wsum1 = 0.f;
extern "C" int block_pack(const float* w, int which, uint16_t* out) {       // w: [128][288] packed fp32 (rows >= 32 zero)
  const float scale = ut::split_weight_scale(w, (size_t)128 * 288);
  g_unscale[which] = 1.f / scale;
  if (which == 0) {
    double ws = 0;
    for (int o = 0; o < 32; ++o) { double rs = 0; for (int k = 0; k < 288; ++k) rs += fabs((double)w[o * 288 + k]); ws = rs > ws ? rs : ws; }
    g_wsum1 = (float)(ws * 1.0001);
  }
  return (int)(ut::pack_split_weights(w, 128, 288, scale, out) != (size_t)2 * 128 * 288);
}

#ifdef BLOCK_W4
namespace ut { hipError_t launch_conv_block32w(const BlockLaunch& b, hipStream_t s); }
#endif

static int* g_dbg = nullptr;            // stamp buffer of the -DW4_STAMPS build (8 x u64 per workgroup)
extern "C" int block_stamps(unsigned long long* host) {      // copies the last w4 launch's stamps (256 x 8 u64)
  if (!g_dbg) return 1;
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(host, g_dbg, 256 * 8 * 8, hipMemcpyDeviceToHost);
}

extern "C" int block_run(const float* in, float* out, const void* w1s, const void* w2s, const float* b1, const float* b2,
                         float bmax1, const unsigned* in_max, int n_img, int hw, int four_waves) {
  ut::BlockLaunch b{};
  b.in = in; b.out = out; b.w1_split = w1s; b.w2_split = w2s; b.unscale_w1 = g_unscale[0]; b.unscale_w2 = g_unscale[1];
  b.bias1 = b1; b.bias2 = b2; b.wsum1 = g_wsum1; b.bmax1 = bmax1; b.in_max = in_max; b.out_max = nullptr; b.status = nullptr;
  b.n_img = n_img; b.H = hw; b.W = hw; b.device = 0; b.num_cu = 256;
  static unsigned* cnt = nullptr;
  if (!cnt) (void)hipMalloc((void**)&cnt, 4);
  (void)hipMemsetAsync(cnt, 0, 4, 0);
  b.tile_counter = cnt;
#ifdef BLOCK_W4
  if (four_waves) {
    if (!g_dbg) { (void)hipMalloc((void**)&g_dbg, 256 * 8 * 8); (void)hipMemset(g_dbg, 0, 256 * 8 * 8); }
    b.status = g_dbg;
    return (int)ut::launch_conv_block32w(b, 0);
  }
#endif
  return (int)ut::launch_conv_block32(b, 0);
}
