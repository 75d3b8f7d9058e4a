// Stem of the backbone: Conv2d(1,32,3,pad=1,bias) + BatchNorm2d + ReLU + MaxPool2d(2)
// (lib/models/model_utils.py:119-124), BatchNorm folded into the 32x9 weights.
// cin = 1, K = 9: not an MFMA shape - direct convolution on the vector ALU.
// One workgroup = one crop x 4 pooled rows; the 10 x 96 input rows it needs are staged in LDS
// with a zero halo.  One thread = one pooled pixel x 4 output channels (a 16-byte NHWC store;
// 8 threads cover the 32 channels = one 128-byte line per pixel).
// The crops arrive as fp32 in [0,1] (the public entry, like the reference's tensor) or as the resampler's exact u8
// grey levels (the fused resample -> backbone path: 4x fewer bytes between the two kernels); (float)u8 / 255.0f is
// the same value the resampler's fp32 output holds, bit for bit.
#include "ut_kernels.h"

namespace ut {

constexpr int CROP = 96, POOLED = 48, STEM_C = 32;
constexpr int ROWS_PER_WG = 4;                 // pooled rows per workgroup
constexpr int IN_ROWS = 2 * ROWS_PER_WG + 2;   // input rows incl. halo
constexpr int IN_COLS = CROP + 2;

__device__ __forceinline__ float stem_px(float v) { return v; }
__device__ __forceinline__ float stem_px(uint8_t v) { return (float)v / 255.0f; }

template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const T* __restrict__ crops,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ out, int n, unsigned* out_max) {
  __shared__ __attribute__((aligned(8))) float tile[IN_ROWS][IN_COLS + 2];   // row stride 100 floats: 8-byte pairs stay aligned
  __shared__ float ws[STEM_C * 9 + STEM_C];
  const int img = blockIdx.y;
  const int prow0 = blockIdx.x * ROWS_PER_WG;
  const T* src = crops + (size_t)img * CROP * CROP;
  for (int i = threadIdx.x; i < STEM_C * 9 + STEM_C; i += 256) ws[i] = i < STEM_C * 9 ? w[i] : bias[i - STEM_C * 9];
  for (int i = threadIdx.x; i < IN_ROWS * IN_COLS; i += 256) {
    int r = i / IN_COLS, c = i - r * IN_COLS;
    int y = 2 * prow0 - 1 + r, x = c - 1;
    tile[r][c] = (y >= 0 && y < CROP && x >= 0 && x < CROP) ? stem_px(src[y * CROP + x]) : 0.f;
  }
  __syncthreads();
  // 4 rows x 48 pooled pixels x 8 channel groups = 1536 work items, 6 per thread.  A thread's channel group
  // (item & 7 == tid & 7) is the same for all of them: its 4 x 9 weights and 4 biases live in registers, and
  // the 4x4 input window is read as 8-byte pairs - the kernel is bound by VALU/LDS issue, not by HBM.
  const int cg = threadIdx.x & 7;
  float wr[4][9], br[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[k][t] = ws[(cg * 4 + k) * 9 + t];
    br[k] = ws[STEM_C * 9 + cg * 4 + k];
  }
  unsigned mx = 0;          // bits of this thread's largest output (post-ReLU: non-negative)
  for (int item = threadIdx.x; item < ROWS_PER_WG * POOLED * 8; item += 256) {
    const int pix = item >> 3;
    const int pr = pix / POOLED, pc = pix - pr * POOLED;
    float in[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float2 lo = *reinterpret_cast<const float2*>(&tile[2 * pr + a][2 * pc]);
      const float2 hi = *reinterpret_cast<const float2*>(&tile[2 * pr + a][2 * pc + 2]);
      in[a][0] = lo.x; in[a][1] = lo.y; in[a][2] = hi.x; in[a][3] = hi.y;
    }
    float res[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float best = -3.4e38f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float acc = 0.f;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc = fmaf(in[dy + ky][dx + kx], wr[k][ky * 3 + kx], acc);
          best = fmaxf(best, acc);
        }
      res[k] = fmaxf(best + br[k], 0.f);
    }
    float* o = out + (((size_t)img * POOLED + prow0 + pr) * POOLED + pc) * STEM_C + cg * 4;
    *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
    const unsigned m01 = max(abs_bits(res[0]), abs_bits(res[1])), m23 = max(abs_bits(res[2]), abs_bits(res[3]));
    mx = max(mx, max(m01, m23));
  }
  // the first layer1 convolution may split these activations into fp16 pieces: it scales them by this word (ut_kernels.h)
  if (out_max) publish_abs_max(out_max, mx);
}

template <typename T>
static hipError_t launch_stem_t(const T* crops, const float* w, const float* bias, float* out, int n, unsigned* out_max,
                                hipStream_t s) {
  if (n <= 0) return hipSuccess;
  // grid.y is limited to 65535: split very large batches
  for (int done = 0; done < n;) {
    int cnt = n - done < 32768 ? n - done : 32768;
    hipLaunchKernelGGL(stem_kernel<T>, dim3(POOLED / ROWS_PER_WG, cnt), dim3(256), 0, s,
                       crops + (size_t)done * CROP * CROP, w, bias,
                       out + (size_t)done * POOLED * POOLED * STEM_C, cnt, out_max);
    done += cnt;
  }
  return hipGetLastError();
}

hipError_t launch_stem(const float* crops, const float* w, const float* bias, float* out, int n, unsigned* out_max,
                       hipStream_t s) {
  return launch_stem_t(crops, w, bias, out, n, out_max, s);
}
hipError_t launch_stem_u8(const uint8_t* crops, const float* w, const float* bias, float* out, int n, unsigned* out_max,
                          hipStream_t s) {
  return launch_stem_t(crops, w, bias, out, n, out_max, s);
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
hipError_t launch_zero_words(void* p, size_t n_words, hipStream_t s) {
  if (!n_words) return hipSuccess;
  size_t blocks = (n_words + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned*)p, n_words);
  return hipGetLastError();
}

__global__ void merge_max_kernel(unsigned* dst, const unsigned* src) { atomicMax(dst, *src); }

hipError_t launch_merge_max(unsigned* dst, const unsigned* src, hipStream_t s) {
  hipLaunchKernelGGL(merge_max_kernel, dim3(1), dim3(1), 0, s, dst, src);
  return hipGetLastError();
}

__global__ void raise_words_kernel(unsigned* w, int n, int add_exp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned v = w[i];
  int e = (int)(v >> 23) & 255;
  if (v == 0u || e == 255) return;
  e = e + add_exp > 254 ? 254 : e + add_exp;
  w[i] = ((unsigned)e << 23) | (v & 0x7FFFFFu);
}

hipError_t launch_raise_words(unsigned* words, int n, int add_exp, hipStream_t s) {
  hipLaunchKernelGGL(raise_words_kernel, dim3((n + 63) / 64), dim3(64), 0, s, words, n, add_exp);
  return hipGetLastError();
}

// Built-in calibration crops.  Crop i is one of four families (i & 3): 0 full-range noise, 1 low-contrast noise around a grey
// level, 2 a linear ramp plus noise, 3 bright Gaussian blobs on a dark ground plus noise (an IR hand image's statistics);
// amplitude and geometry vary with i >> 2.  Values are grey levels k / 255 like lib/tracker/tracker.py:332 produces.
__device__ __forceinline__ unsigned calib_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void calibration_crops_kernel(float* out, int n) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * CROP * CROP) return;
  const int i = idx / (CROP * CROP), px = idx % (CROP * CROP), y = px / CROP, x = px % CROP;
  const unsigned hp = calib_hash((unsigned)idx * 2654435761u + 12345u);
  const float u = (float)(hp >> 8) * (1.0f / 16777216.0f);                       // pixel noise in [0, 1)
  const unsigned hc = calib_hash((unsigned)i * 40503u + 977u);
  const float a = (float)(hc & 255) * (1.0f / 255.0f), b = (float)((hc >> 8) & 255) * (1.0f / 255.0f),
              c = (float)((hc >> 16) & 255) * (1.0f / 255.0f);
  const float fx = (float)x * (1.0f / CROP), fy = (float)y * (1.0f / CROP);
  float v;
  switch (i & 3) {
    case 0: v = u; break;
    case 1: v = 0.1f + 0.8f * a + (0.02f + 0.2f * b) * (u - 0.5f); break;
    case 2: v = a * fx + b * fy + 0.3f * c * u; break;
    default: {
      const float dx0 = fx - a, dy0 = fy - b, dx1 = fx - c, dy1 = fy - a;
      const float s0 = 0.05f + 0.2f * c, s1 = 0.05f + 0.2f * b;
      v = 0.05f + 0.9f * __expf(-(dx0 * dx0 + dy0 * dy0) / (2.f * s0 * s0)) + 0.6f * __expf(-(dx1 * dx1 + dy1 * dy1) / (2.f * s1 * s1)) + 0.1f * u;
    }
  }
  v = fminf(fmaxf(v, 0.f), 1.f);
  out[idx] = floorf(v * 255.f + 0.5f) / 255.0f;
}

hipError_t launch_calibration_crops(float* crops, int n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(calibration_crops_kernel, dim3((n * CROP * CROP + 255) / 256), dim3(256), 0, s, crops, n);
  return hipGetLastError();
}

}  // namespace ut
