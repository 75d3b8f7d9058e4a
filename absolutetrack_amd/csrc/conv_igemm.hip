// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces every Conv2d(+BatchNorm2d)(+residual)(+ReLU) of the reference after the stem:
// lib/models/backbone_resnet.py:56-72 (BasicBlock), lib/models/model_utils.py:134 (projection),
// :141-163 (fusion), lib/models/temporal.py:31-38, lib/models/model_utils.py:195-208 (regressor).
//
// GEMM view: M = n_img*Ho*Wo output pixels, N = cout, K = taps*cin, k ordered (channel slice, tap,
// channel) - see ut_kernels.h.
// Activations are NHWC so a k-run of 4 channels is one 16-byte load; weights are pre-packed
// [cout_pad][k_pad] (k contiguous) with BatchNorm folded in.  A workgroup (4 waves, 256 threads)
// owns a BM x BN output tile and walks K in chunks of 32:
//   global (im2col gather, zero fill at the borders) -> registers -> LDS (double buffered,
//   rows padded to 36 floats so ds_read_b128 fragment reads are bank-conflict free)
//   -> one ds_read_b128 per 32-row fragment per 8 k  -> 4 MFMA 32x32x2 per fragment pair.
// Lane l of a wave holds row (l&31) of the fragment and k-half (l>>5); the 4 floats of a b128
// read feed 4 consecutive MFMAs (k order inside the 8-run is permuted identically for A and B).
// Epilogue: bias (+residual) (+ReLU) from the accumulator layout col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "ut_kernels.h"

namespace ut {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_ROW = BK + 4;   // floats; 144 B row stride = 9 x 16 B -> conflict-free b128 reads

template <int BM, int BN, int WR, int WC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvLaunch p) {
  static_assert(WR * WC == 4, "4 waves per workgroup");
  constexpr int MI = BM / WR / 32;   // 32x32 accumulator tiles per wave along M
  constexpr int NI = BN / WC / 32;   // ... along N
  constexpr int AP = BM / 32;        // 16-byte loads per thread per chunk for the A tile
  constexpr int BP = BN / 32;
  constexpr int STAGE = (BM + BN) * LDS_ROW;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WC, wn = wave % WC;
  const int g = tid & 7;        // which 4-float group of the 32-wide k chunk this thread stages
  const int r0 = tid >> 3;      // first tile row this thread stages (then +32 per pass)

  const int M = p.n_img * p.Ho * p.Wo;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread im2col row bookkeeping (fixed over the K loop)
  // Activations and weights are read through raw buffer descriptors: a tap that falls outside the
  // image (or a row beyond M) gets an out-of-range offset and the hardware returns zeros - no
  // branches and no selects on the load path, so the loads stay in flight under the MFMAs.
  // k beyond taps*cin needs no masking: the packed weights are zero there and every address that
  // passes the bounds test holds a finite activation.
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in), 0, (int)((size_t)p.n_img * p.H * p.W * p.cin * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)((size_t)p.cout_pad * p.k_pad * sizeof(float)), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  int a_pix[AP];                 // element offset of the (iy0, ix0) pixel of this row (may be "negative")
  int a_iy[AP], a_ix[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    int m = m0 + r0 + 32 * i;
    bool ok = m < M;
    int mm = ok ? m : 0;
    int img = mm / (p.Ho * p.Wo);
    int rem = mm - img * (p.Ho * p.Wo);
    int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    a_iy[i] = ok ? oy * p.stride - p.pad : -100000;   // rows beyond M never pass the bounds test
    a_ix[i] = ox * p.stride - p.pad;
    a_pix[i] = ((img * p.H + a_iy[i]) * p.W + a_ix[i]) * p.cin;
  }
  unsigned b_off = (unsigned)(((n0 + r0) * p.k_pad + 4 * g) * 4);
  const unsigned b_row_step = (unsigned)(32 * p.k_pad * 4);

  // (slice, tap, channel in slice) of this thread's 4-float group; cslice >= 32, so one step of 32
  // crosses at most one tap boundary, and taps wrap into the next channel slice
  const int taps = p.ksize * p.ksize;
  int tap = (4 * g) / p.cslice;
  int ch = 4 * g - tap * p.cslice;
  int ch_base = 0;                // first channel of the current slice
  if (tap >= taps) { tap -= taps; ch_base = p.cslice; }

  u32x4 a_reg[AP], b_reg[BP];

#define UT_FETCH()                                                                                   \
  {                                                                                                  \
    int dy = 0, dx = 0;                                                                              \
    if (p.ksize == 3) { dy = (tap * 11) >> 5; dx = tap - 3 * dy; } /* tap/3 for tap < 32 */         \
    const int tap_off = (dy * p.W + dx) * p.cin + ch_base + ch;                                      \
    _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                                 \
      const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;                                                \
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;                  \
      const unsigned off = ok ? (unsigned)(a_pix[i] + tap_off) * 4u : OOB;                           \
      a_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0);                           \
    }                                                                                                \
    _Pragma("unroll") for (int i = 0; i < BP; ++i)                                                   \
      b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_off + i * b_row_step, 0, 0);        \
    b_off += BK * 4;                                                                                 \
    ch += BK;                                                                                        \
    if (ch >= p.cslice) { ch -= p.cslice; ++tap; }                                                   \
    if (tap >= taps) { tap -= taps; ch_base += p.cslice; }                                           \
  }

#define UT_STAGE(buf)                                                                                \
  {                                                                                                  \
    float* as_ = smem + (buf) * STAGE;                                                               \
    float* bs_ = as_ + BM * LDS_ROW;                                                                 \
    _Pragma("unroll") for (int i = 0; i < AP; ++i)                                                   \
      *reinterpret_cast<u32x4*>(as_ + (r0 + 32 * i) * LDS_ROW + 4 * g) = a_reg[i];                   \
    _Pragma("unroll") for (int i = 0; i < BP; ++i)                                                   \
      *reinterpret_cast<u32x4*>(bs_ + (r0 + 32 * i) * LDS_ROW + 4 * g) = b_reg[i];                   \
  }

  const int fr = lane & 31;          // fragment row (A/B) == accumulator column
  const int fh = lane >> 5;          // k half (A/B) == accumulator row offset 4*fh
  const int hw = p.Ho * p.Wo;

  // accumulators start at bias (+ residual): the residual tile is fetched here, under the first
  // im2col fetch, instead of in a serialised load->add->store epilogue
  f32x16 acc[MI][NI];
  {
    // the residual goes through a buffer descriptor as well: with no residual the descriptor is
    // empty and every load returns 0 - one straight-line burst of loads, no per-element branches
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.res ? p.res : p.bias), 0,
        p.res ? (int)((size_t)M * p.cout_store * sizeof(float)) : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * (NI * 32) + j * 32 + fr;
      const float bias = p.bias[n];   // bias is padded to cout_pad
      const bool n_ok = n < p.cout_store;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * (MI * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          const unsigned off = (n_ok && m < M) ? (unsigned)(m * p.cout_store + n) * 4u : OOB;
          acc[i][j][e] = bias + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off, 0, 0));
        }
    }
  }

  const int n_chunks = p.k_pad / BK;

  UT_FETCH();
  UT_STAGE(0);
  __syncthreads();

  for (int c = 0; c < n_chunks; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < n_chunks;
    if (more) UT_FETCH();
    const float* as = smem + buf * STAGE + (wm * (MI * 32) + fr) * LDS_ROW + 4 * fh;
    const float* bs = smem + buf * STAGE + BM * LDS_ROW + (wn * (NI * 32) + fr) * LDS_ROW + 4 * fh;
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      float4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_ROW + 8 * q);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_ROW + 8 * q);
      // the next chunk goes to the other LDS buffer while this chunk's last MFMAs are still queued:
      // the wave reaches the barrier with the matrix pipe busy instead of draining it first
      if (q == BK / 8 - 1 && more) UT_STAGE(buf ^ 1);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
#undef UT_FETCH
#undef UT_STAGE

  // ---- epilogue: (ReLU) and store; residual layout == output layout, NHWC only
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn * (NI * 32) + j * 32 + fr;
    const bool n_ok = n < p.cout_store;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (MI * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (n_ok && m < M) {
          float v = acc[i][j][e];
          size_t o;
          if (p.out_nchw) {
            int img = m / hw;
            o = ((size_t)img * p.cout_store + n) * hw + (m - img * hw);
          } else {
            o = (size_t)m * p.cout_store + n;
          }
          if (p.relu) v = fmaxf(v, 0.f);
          p.out[o] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WR, int WC>
static hipError_t launch_cfg(const ConvLaunch& c, hipStream_t s) {
  const int M = c.n_img * c.Ho * c.Wo;
  dim3 grid((M + BM - 1) / BM, (c.cout_store + BN - 1) / BN);
  size_t lds = 2 * (size_t)(BM + BN) * LDS_ROW * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WR, WC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WR, WC>), grid, dim3(256), lds, s, c);
  return hipGetLastError();
}

hipError_t launch_conv_igemm(const ConvLaunch& c, hipStream_t s) {
  if (c.res && c.out_nchw) return hipErrorInvalidValue;
  if (c.cin % 4 != 0 || c.cin < BK || c.k_pad % BK != 0 || c.cout_pad % 128 != 0 || c.ksize * c.ksize > 9)
    return hipErrorInvalidValue;
  // 32-bit byte offsets into the activation / residual tensors
  if ((size_t)c.n_img * c.Ho * c.Wo * c.cout_store * sizeof(float) >= 0x7FFFFF00ull) return hipErrorInvalidValue;
  if ((size_t)c.n_img * c.H * c.W * c.cin * sizeof(float) >= 0x7FFFFF00ull) return hipErrorInvalidValue;
  if (c.cout_store <= 32) return launch_cfg<128, 32, 4, 1>(c, s);
  if (c.cout_store <= 64) return launch_cfg<128, 64, 2, 2>(c, s);
  return launch_cfg<128, 128, 2, 2>(c, s);
}

}  // namespace ut
