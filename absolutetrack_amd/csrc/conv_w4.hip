// 3x3 / stride-1 convolutions from 64 input channels up to a multiple of 128 output channels (layer3's nine 128 -> 128 and
// layer4's three 256 -> 256 convolutions, lib/models/backbone_resnet.py:56-72 at 12x12x128 and 6x6x256: 44 % of a step) in the
// split-fp16 arithmetic of conv_split.hip, as FOUR waves - one per SIMD, each with the SIMD's whole 512-entry register file - with
// NO synchronisation inside a 32-channel slice and NO address arithmetic in the MFMA stream.
//
// conv_split_kernel<256, 128, 4, 2, true> (eight waves of 64 x 64) pays ~1,600 cycles per 32-deep chunk on top of its MFMAs
// whatever their number: every chunk's weights go global -> LDS and are published to all eight waves through a counter, a wave's
// 12 MFMAs per k-step have 8 fragment reads to wait for, and every fragment read computes a swizzled, masked address.  Here
//   * a wave owns 128 pixels x 64 output channels of the 256 x 128 tile: 24 MFMAs per k-step on 8 pixel-fragment reads;
//   * the weights never touch LDS: a wave loads the fragments of ITS 64 output channels straight from the fragment-ordered
//     planes into registers (one coalesced 1-KB load per fragment, W4_DIST k-steps ahead; the two waves of a channel half hit
//     in L1 behind each other) - no weight ring, no counter, nothing to publish;
//   * the input patch of a slice lives in LDS in PADDED image coordinates: pixel (img, y, x) of a W x H map at row
//     img (H + 1)(W + 1) + y (W + 1) + x, the rows in between (one per image row, W + 1 per image) hold zeros.  A tap is then a
//     CONSTANT row shift - out-of-image taps land on the zero rows, no mask, no select - and the 16-byte groups of a row are
//     W4_CAP rows apart (consecutive rows = consecutive 16-byte words: conflict-free without a swizzle), so a fragment read is
//     one ds_read_b128 of a per-lane base register with everything else in the instruction's immediate offset;
//   * the patch of the next slice comes global -> registers -> LDS: every thread loads ten float4, splits them into the two
//     fp16 pieces and stores the pieces at their padded position - one load, one conversion and two 8-byte LDS stores per k-step,
//     between the MFMAs; no LDS-DMA, no conversion pass;
//   * ONE barrier per slice (every 432 MFMAs of a wave): behind it the patch just written is read, the one just read rewritten;
//   * the epilogue requests a wave's 128 residual values in one go (one memory latency per tile) through two per-lane offset
//     registers and scalar row offsets.
// Every vector-memory operation is a compiler-visible load or store, so the waits are the compiler's; the order of a k-step's
// instructions is pinned slot by slot (one MFMA per slot).
// Same tensors, same weight planes and - per output element - the same products in the same order as
// conv_split_kernel<256, 128, 4, 2, true>: bit-identical results (tests/test_gpu_parity.py::test_split_f16_four_wave_kernel_...).
#include <atomic>

#include "ut_kernels.h"

namespace ut {
namespace {

typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));

constexpr int W4_BM = 256, W4_BN = 128;
constexpr int W4_MI = 4, W4_NI = 2;            // 32 x 32 blocks of a wave: 128 pixels x 64 channels
constexpr int W4_CAP = 384;                    // padded patch rows of a slice (256 pixels + their zero rows + a halo of W + 2 either side)
constexpr int W4_GSTRIDE = W4_CAP * 16;        // bytes between two 16-byte groups of a row: groups 0..3 first pieces, 4..7 remainders
constexpr int W4_STAGE = 8 * W4_GSTRIDE;       // one slice patch: 48 KB
constexpr int W4_SLOT = 2 * W4_STAGE;          // the next tile's index
constexpr int W4_LDS = W4_SLOT + 16;
constexpr int W4_LROWS = 320;                  // pixel rows loaded per slice: the tile's 256 and 32 either side (>= 2 W + 3)
constexpr int W4_NLOAD = W4_LROWS * 8 / 256;   // float4 patch loads per thread and slice: 10
constexpr int W4_NSTG = 5;                     // patch loads in flight per thread (loaded in k-step q, split in k-step q + 4)
constexpr int W4_DIST = 2;                     // weight fragments are loaded this many k-steps ahead (in-order vmcnt: a wait for
constexpr int W4_NSET = W4_DIST + 1;           // them also waits for every older load and store; 18 k-steps % W4_NSET == 0)
constexpr unsigned W4_HOOB = 0x80000000u;      // out-of-range offset that stays out of range with a scalar offset added
static_assert(18 % W4_NSET == 0, "the register sets of the weight fragments line up across slices");

// the pieces of a * s and b * s for a power of two s (conv_split.hip::split_pair_scaled)
__device__ __forceinline__ void w4_split_pair(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2w h = __builtin_bit_cast(f16x2w, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
// padded row of pixel m (any integer: pixels in front of the tensor continue the pattern), and its position in the image
template <int WI, int HI>
__device__ __forceinline__ int w4_ppos(int m, int& x, int& y) {
  constexpr int HW = WI * HI;
  const int img = m >= 0 ? m / HW : -((HW - 1 - m) / HW);
  const int rem = m - img * HW;
  y = rem / WI;
  x = rem - y * WI;
  return img * ((HI + 1) * (WI + 1)) + y * (WI + 1) + x;
}

template <int WI, int HI>
__global__ __launch_bounds__(256, 1) void conv_w4_kernel(ConvLaunch p, int tiles_n, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MI = W4_MI, NI = W4_NI, BM = W4_BM, BN = W4_BN, PW = WI + 1;
  static_assert(255 + (255 / WI + 2) + (255 / (WI * HI) + 2) * PW + 2 * (PW + 1) < W4_CAP - 1, "the padded tile fits the patch buffer in front of its last row");
  static_assert((W4_LROWS - BM) / 2 >= 2 * WI + 3, "the loaded rows reach every zero row's owner");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;

  const int M = p.n_img * WI * HI;
  const int n_slices = p.cin / 32;
  const int n_chunks = p.k_pad / 32;           // 9 * n_slices

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in), 0, (int)((size_t)M * p.cin * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.w_split), 0, (int)((size_t)(p.cout_pad / 32) * n_chunks * 4096), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.bias), 0, p.res ? (int)((size_t)M * p.cout_store * sizeof(float)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)M * p.cout_store * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.tile_counter, 0, 4, 0x00020000);

  float x_scale = 1.f, x_unscale = 1.f;
  if (p.in_max) {
    bool ok;
    split_act_scale(p.in_max, p.in_obs, x_scale, x_unscale, ok);
    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);
  }
  const float tot_unscale = p.split_unscale * x_unscale;
  const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();
  const unsigned row_b = (unsigned)p.cout_store * 4u;

  const int grid = gridDim.x;
  int slot = blockIdx.x;
  if ((grid & 7) == 0) slot = (blockIdx.x & 7) * (grid >> 3) + (blockIdx.x >> 3);      // first round XCD-contiguous

  // ---- patch stream: thread t loads group t & 7 (four channels) of pixel rows m0 - 32 + 32 j + (t >> 3), j = 0 .. 9, of the tile
  // being fetched.  h_base: the per-lane byte offset of row t >> 3 (rows in front of the tensor wrap far beyond 2^31, rows behind
  // it are behind the descriptor's range: both load zeros with no compare).  h_wa[j]: where the row's first pieces go in the patch
  // buffer (group (t & 7) >> 1, half t & 1; the remainders four groups on); rows without a place go to the buffer's spare last row; bit j of
  // h_xe / h_ye: it ends an image row / sits in an image's last row - it then owns the zero row(s) behind / below it.
  unsigned h_base = 0;
  const unsigned h_step = (unsigned)(32 * p.cin) * 4u;
  unsigned h_wa[W4_NLOAD];
  unsigned h_own0 = 0;            // row 0's own place even when it lies in front of the buffer (it may own a zero row inside)
  unsigned h_xe = 0, h_ye = 0, h_xc = 0;      // (h_xc: the row ends an image row, whatever its place)
#define W4_H_SETUP(TILE)                                                                             \
  {                                                                                                  \
    const int m0_ = ((TILE) / tiles_n) * BM;                                                         \
    int x0_, y0_;                                                                                    \
    const int lo_ = w4_ppos<WI, HI>(m0_, x0_, y0_) - PW - 1;                                         \
    h_base = (unsigned)((m0_ - 32 + (tid >> 3)) * p.cin + 4 * (tid & 7)) * 4u;                       \
    h_xe = h_ye = h_xc = 0;                                                                            \
    /* row j is 32 pixels behind row j - 1: its place in the image follows by carries, not by another division */ \
    int x_, y_;                                                                                      \
    int rel_ = w4_ppos<WI, HI>(m0_ - 32 + (tid >> 3), x_, y_) - lo_;                                 \
    _Pragma("unroll") for (int j = 0; j < W4_NLOAD; ++j) {                                           \
      if (j > 0) {                                                                                   \
        x_ += 32 % WI; y_ += 32 / WI; rel_ += (32 / WI) * PW + 32 % WI;                              \
        if (x_ >= WI) { x_ -= WI; y_ += 1; rel_ += 1; }                                              \
        if (y_ >= HI) { y_ -= HI; rel_ += PW; }                                                      \
        if (y_ >= HI) { y_ -= HI; rel_ += PW; }                                                      \
      }                                                                                              \
      const bool ok_ = rel_ >= 0 && rel_ < W4_CAP;                                                   \
      /* (a row without a place goes to the buffer's last row, which no tile reaches: the stores need no predicate) */ \
      h_wa[j] = (unsigned)((ok_ ? rel_ : W4_CAP - 1) * 16 + ((tid & 7) >> 1) * W4_GSTRIDE + (tid & 1) * 8); \
      if (j == 0) h_own0 = (unsigned)(rel_ * 16 + ((tid & 7) >> 1) * W4_GSTRIDE + (tid & 1) * 8);      \
      /* (a zero row inside the buffer is written by its owner even when the owner itself lies in front of the buffer) */ \
      h_xe |= (x_ == WI - 1 && rel_ + 1 >= 0 && rel_ + 1 < W4_CAP ? 1u : 0u) << j;                   \
      h_ye |= (y_ == HI - 1 && rel_ + PW >= 0 && rel_ + PW + 1 < W4_CAP ? 1u : 0u) << j;             \
      h_xc |= (x_ == WI - 1 ? 1u : 0u) << j;                                                         \
    }                                                                                                \
  }
#define W4_H_OFF(J) (h_base + (unsigned)(J) * h_step)
  // the zero rows of the layout of the tile being fetched, in patch buffer WB (this thread's half of one first-piece group and of
  // its remainder group): behind a row that ends an image row, below a row of an image's last row, and the corner between them
#define W4_H_ZERO(WB)                                                                                \
  {                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < W4_NLOAD; ++j) {                                           \
      const unsigned a_ = (WB) + (j == 0 ? h_own0 : h_wa[j]);                                        \
      const bool xe_ = (h_xe >> j) & 1u, ye_ = (h_ye >> j) & 1u;                                     \
      if (xe_) {                                                                                     \
        *reinterpret_cast<u32x2w*>(smem + a_ + 16u) = u32x2w{0, 0};                                  \
        *reinterpret_cast<u32x2w*>(smem + a_ + 16u + 4 * W4_GSTRIDE) = u32x2w{0, 0};                 \
      }                                                                                              \
      if (ye_) {                                                                                     \
        *reinterpret_cast<u32x2w*>(smem + a_ + PW * 16u) = u32x2w{0, 0};                             \
        *reinterpret_cast<u32x2w*>(smem + a_ + PW * 16u + 4 * W4_GSTRIDE) = u32x2w{0, 0};            \
      }                                                                                              \
      if (ye_ && ((h_xc >> j) & 1u)) {                                                               \
        *reinterpret_cast<u32x2w*>(smem + a_ + (PW + 1) * 16u) = u32x2w{0, 0};                       \
        *reinterpret_cast<u32x2w*>(smem + a_ + (PW + 1) * 16u + 4 * W4_GSTRIDE) = u32x2w{0, 0};      \
      }                                                                                              \
    }                                                                                                \
  }
  // ---- read side: per-lane byte offset of the lane's pixel of block i (its padded row minus the tile's first, + the lane's
  // k half as a group); taps, k-step halves and pieces are immediates
  unsigned pb[MI];
#define W4_P_SETUP(TILE)                                                                             \
  {                                                                                                  \
    const int m0_ = ((TILE) / tiles_n) * BM;                                                         \
    int x_, y_;                                                                                      \
    const int p0_ = w4_ppos<WI, HI>(m0_, x_, y_);                                                    \
    int rel_ = w4_ppos<WI, HI>(m0_ + wm * (MI * 32) + fr, x_, y_) - p0_;                             \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                 \
      if (i > 0) {                                                                                   \
        x_ += 32 % WI; y_ += 32 / WI; rel_ += (32 / WI) * PW + 32 % WI;                              \
        if (x_ >= WI) { x_ -= WI; y_ += 1; rel_ += 1; }                                              \
        if (y_ >= HI) { y_ -= HI; rel_ += PW; }                                                      \
        if (y_ >= HI) { y_ -= HI; rel_ += PW; }                                                      \
      }                                                                                              \
      pb[i] = (unsigned)(rel_ * 16 + fh * W4_GSTRIDE);                                               \
    }                                                                                                \
  }

  f32x16w acc[MI][NI];
  u32x4w xp[MI][2];               // pixel fragments (first piece, remainder): ONE set - a fragment of the next k-step is read into
                                  // its registers as soon as this k-step's last MFMA on it has been issued
  u32x4w wf[W4_NSET][NI][2];      // weight fragments (plane 0, plane 1)
  float4 stg[W4_NSTG];            // patch values between their load and their split
  const unsigned w_lane = (unsigned)lane * 16u;

  // pixel fragment PC (0 first piece, 1 remainder) of block I for k-step half S of tap TAP, base register BASE (buffer included)
#define W4_READ_X(I, PC, S, TAP, BASE)                                                               \
  xp[I][PC] = *reinterpret_cast<const u32x4w*>(smem + (BASE)[I] + (unsigned)(((PC) * 4 + 2 * (S)) * W4_GSTRIDE + (((TAP) / 3) * PW + (TAP) % 3) * 16));
  // weight fragment IDX (block IDX / 2, plane IDX % 2) of chunk CH, k-step half S, of the tile column at byte offset WROW
#define W4_LOAD_W(SET, IDX, CH, S, WROW)                                                             \
  {                                                                                                  \
    const unsigned so_ = (WROW) + (unsigned)(((wn * NI + (IDX) / 2) * n_chunks + (CH)) * 4096 + ((S) * 2 + (IDX) % 2) * 1024); \
    wf[SET][(IDX) / 2][(IDX) % 2] = __builtin_bit_cast(u32x4w, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, so_, 0)); \
  }
#define W4_MFMA(WSET, N)                                                                             \
  {                                                                                                  \
    constexpr int pr_ = (N) / (MI * NI), ij_ = (N) % (MI * NI), i_ = ij_ / NI, j_ = ij_ % NI;        \
    constexpr int wp_ = pr_ == 1 ? 1 : 0, xq_ = pr_ == 0 ? 1 : 0;      /* small terms first: x1 w0, x0 w1, x0 w0 */ \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, xp[i_][xq_]),                   \
                                                         __builtin_bit_cast(f16x8w, wf[WSET][j_][wp_]), acc[i_][j_], 0, 0, 0); \
  }
  // (the empty asm keeps memory operations, the scheduling barriers everything else, inside their slot)
#define W4_PIN() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define W4_ROW_OFF(I, R) ((unsigned)((I) * 32 + 8 * ((R) >> 2) + ((R) & 3)) * row_bs)
  // split the patch value loaded as J and store its pieces at their padded place in buffer WB (rows without one: the spare last row)
#define W4_STORE_PATCH(V, J, WB)                                                                     \
  {                                                                                                  \
    unsigned a0_, b0_, a1_, b1_;                                                                     \
    w4_split_pair((V).x, (V).y, x_scale, a0_, b0_);                                                  \
    w4_split_pair((V).z, (V).w, x_scale, a1_, b1_);                                                  \
    *reinterpret_cast<u32x2w*>(smem + (WB) + h_wa[J]) = u32x2w{a0_, a1_};                            \
    *reinterpret_cast<u32x2w*>(smem + (WB) + h_wa[J] + 4 * W4_GSTRIDE) = u32x2w{b0_, b1_};           \
  }

  int tile = slot;
  int next_tile = 0;
  int cur_buf = 0;                // patch buffer (0 / 1) of the slice being computed
  unsigned out_bits = 0;
  const unsigned slot_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem + (unsigned)W4_SLOT;

  // ---- prologue (exposed once per workgroup): both patch buffers zeroed, the first tile's first patch, the weights of its
  // first W4_DIST k-steps and the pixel fragments of its first
#pragma unroll
  for (int k = 0; k < 2 * W4_STAGE / (256 * 16); ++k) *reinterpret_cast<u32x4w*>(smem + (k * 256 + tid) * 16) = u32x4w{0, 0, 0, 0};
  W4_H_SETUP(tile);
  W4_P_SETUP(tile);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < W4_NLOAD; ++j) {
    const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, W4_H_OFF(j), 0, 0));
    W4_STORE_PATCH(v, j, 0u)
  }
  unsigned w_row = (unsigned)((tile % tiles_n) * (BN / 32)) * (unsigned)n_chunks * 4096u;     // byte offset of the tile column's planes
  unsigned w_row_next = w_row;
#pragma unroll
  for (int g = 0; g < W4_DIST; ++g)
#pragma unroll
    for (int idx = 0; idx < 4; ++idx) { W4_LOAD_W(g, idx, g >> 1, g & 1, w_row) }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    W4_READ_X(i, 1, 0, 0, pb)
    if (i < MI - 1) { W4_READ_X(i, 0, 0, 0, pb) }      // (the last block's first piece: read in slot 0 of its own k-step)
  }

  // One slice: 18 k-steps of 24 slots, one MFMA per slot.  Slot N of k-step q -
  //   0             the first piece of THIS k-step's last pixel block (its registers were busy until the k-step before ended);
  //   2, 4, 6, 8    the remainder pieces of the next k-step's four pixel blocks;   18, 20, 22  the first pieces of its blocks 0 .. 2
  //                 (k-step 17: of the next slice's / tile's first k-step, out of the other patch buffer);
  //   9 .. 12       the weight fragments of k-step q + W4_DIST;
  //   13            a patch load of the next slice (q < 10);   14, 15   the split and store of the patch load of four k-steps ago;
#define W4_SLOT_BODY(N)                                                                         \
        {                                                                                            \
          if ((N) == 0 && q != 17) { W4_READ_X(MI - 1, 0, q & 1, q >> 1, rb) }      /* (k-step 17: in front of the barrier) */ \
          if ((N) >= 2 && (N) <= 8 && ((N) & 1) == 0 && q1 < 18) { W4_READ_X(((N) / 2 - 1) & 3, 1, q1 & 1, q1 >> 1, rb) } \
          if ((N) >= 2 && (N) <= 8 && ((N) & 1) == 0 && q1 == 18) { W4_READ_X(((N) / 2 - 1) & 3, 1, 0, 0, wb) } \
          if ((N) >= 18 && (N) <= 22 && ((N) & 1) == 0 && q1 < 18) { W4_READ_X(((N) / 2 - 9) & 3, 0, q1 & 1, q1 >> 1, rb) } \
          if ((N) >= 18 && (N) <= 22 && ((N) & 1) == 0 && q1 == 18) { W4_READ_X(((N) / 2 - 9) & 3, 0, 0, 0, wb) } \
          if ((N) >= 9 && (N) < 13 && qd < 18) { W4_LOAD_W(qd % W4_NSET, ((N) - 9) & 3, ch0 + (qd >> 1), qd & 1, w_row) } \
          if ((N) >= 9 && (N) < 13 && qd >= 18) { W4_LOAD_W(qd % W4_NSET, ((N) - 9) & 3, ch_after + ((qd - 18) >> 1), qd & 1, row_after) } \
          if ((N) == 13 && q < W4_NLOAD)                                                             \
            stg[q % W4_NSTG] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, W4_H_OFF(q), f_soff, 0)); \
          if ((N) == 10 && q == 10 && sl == 0 && tid == 0) *reinterpret_cast<int*>(smem + W4_SLOT) = grid + ticket; \
          if ((N) == 14 && q >= 4 && q < 4 + W4_NLOAD) { const float4 v_ = stg[(q + W4_NSTG - 4) % W4_NSTG]; w4_split_pair(v_.x, v_.y, x_scale, cv0, cv1); } \
          if ((N) == 15 && q >= 4 && q < 4 + W4_NLOAD) {                                             \
            const float4 v_ = stg[(q + W4_NSTG - 4) % W4_NSTG];                                      \
            unsigned a1_, b1_;                                                                       \
            w4_split_pair(v_.z, v_.w, x_scale, a1_, b1_);                                            \
            *reinterpret_cast<u32x2w*>(smem + wbuf + h_wa[(q + 6) % 10]) = u32x2w{cv0, a1_};         \
            *reinterpret_cast<u32x2w*>(smem + wbuf + h_wa[(q + 6) % 10] + 4 * W4_GSTRIDE) = u32x2w{cv1, b1_}; \
          }                                                                                          \
          W4_PIN();                                                                                  \
          W4_MFMA(ws, N);                                                                            \
          W4_PIN();                                                                                  \
        }
#define W4_SLICE()                                                                                   \
      _Pragma("clang loop unroll(full)") for (int q = 0; q < 18; ++q) {                              \
        const int ws = q % W4_NSET;                                                                  \
        unsigned cv0 = 0, cv1 = 0;     /* the first half of the patch value being split (slot 14 -> 15) */ \
        /* (q + 1): the k-step whose pixel fragments are read now; (q + W4_DIST): the k-step whose weights are loaded now */ \
        const int q1 = q + 1, qd = q + W4_DIST;                                                      \
        if (q == 17) {                                                                               \
          /* every wave has written its part of the next patch (k-steps 4 .. 13) and has read its last fragments of this one \
             (slot 0 of this k-step excepted: see below): ONE barrier per slice, in front of the first reads of the next patch */ \
          W4_READ_X(MI - 1, 0, 1, 8, rb)                                                             \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
          __builtin_amdgcn_s_barrier();                                                              \
          asm volatile("" ::: "memory");                                                             \
          if (sl == 0) {                                                                             \
            int nv;                                                                                  \
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nv) : "v"(slot_addr) : "memory"); \
            next_tile = __builtin_amdgcn_readfirstlane(nv);                                          \
            w_row_next = (unsigned)((next_tile % tiles_n) * (BN / 32)) * (unsigned)n_chunks * 4096u; \
          }                                                                                          \
          if (last_slice) W4_P_SETUP(next_tile);    /* the fragments read in this k-step are the next tile's */ \
          _Pragma("unroll") for (int i = 0; i < MI; ++i) wb[i] = pb[i] + wbuf;                       \
        }                                                                                            \
        W4_SLOT_BODY(0) W4_SLOT_BODY(1) W4_SLOT_BODY(2) W4_SLOT_BODY(3) W4_SLOT_BODY(4) W4_SLOT_BODY(5) \
        W4_SLOT_BODY(6) W4_SLOT_BODY(7) W4_SLOT_BODY(8) W4_SLOT_BODY(9) W4_SLOT_BODY(10) W4_SLOT_BODY(11) \
        W4_SLOT_BODY(12) W4_SLOT_BODY(13) W4_SLOT_BODY(14) W4_SLOT_BODY(15) W4_SLOT_BODY(16) W4_SLOT_BODY(17) \
        W4_SLOT_BODY(18) W4_SLOT_BODY(19) W4_SLOT_BODY(20) W4_SLOT_BODY(21) W4_SLOT_BODY(22) W4_SLOT_BODY(23) \
      }

  for (;;) {
    // the tile after this one: the ticket is taken here, written to LDS by thread 0 in the middle of the tile's first slice and
    // read by everyone behind that slice's barrier (a tile has at least two slices)
    int ticket = 0;
    if (tid == 0) ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int sl = 0; sl < n_slices; ++sl) {
      const bool last_slice = sl == n_slices - 1;
      unsigned rbuf = (unsigned)__builtin_amdgcn_readfirstlane(cur_buf * W4_STAGE), wbuf = (unsigned)__builtin_amdgcn_readfirstlane((cur_buf ^ 1) * W4_STAGE);
      unsigned row_bs = (unsigned)__builtin_amdgcn_readfirstlane((int)row_b);
      asm volatile("" : "+s"(rbuf), "+s"(wbuf), "+s"(row_bs));      // (opaque per slice: nothing derived from them is carried across slices)
      // the patch fetched during this slice: the next slice of this tile, or slice 0 of the next tile - whose layout the buffer
      // gets now: its zero rows are written here (and once more for the other buffer in the next tile's first slice)
      if (last_slice) {
        W4_H_SETUP(next_tile);
        W4_H_ZERO(wbuf)
      } else if (sl == 0) {
        W4_H_ZERO(wbuf)
      }
      const unsigned f_soff = last_slice ? 0u : (unsigned)(sl + 1) * 128u;
      const int ch0 = sl * 9;
      // the weight fragments of the k-steps behind this slice: the next slice's first chunks, or the next tile's
      const int ch_after = last_slice ? 0 : ch0 + 9;
      const unsigned row_after = last_slice ? w_row_next : w_row;
      unsigned rb[MI], wb[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) { rb[i] = pb[i] + rbuf; wb[i] = pb[i] + wbuf; }
      W4_SLICE()
      cur_buf ^= 1;
    }
    // ---- epilogue: 1 / (weight scale x activation scale) x accumulator + bias + residual, ReLU, store.  The pixels are the
    // MFMAs' first operand, so a lane's sixteen registers of a block are ONE output channel of sixteen pixels and one dword access
    // per register covers two whole 128-byte half rows (conv_split.hip); a block's sixteen rows are one per-lane base plus
    // WAVE-UNIFORM row offsets in the instruction's scalar offset (two offset registers per wave, not one per access).  All 128
    // residual requests of a wave go out before the first value is needed: one memory latency per tile, not one per block.  The
    // scalar offset is not part of the descriptor's range check: the tile that reaches beyond the tensor takes per-access offsets.
    {
      const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
      const bool ragged = tm * BM + BM > M;
      float bb[NI];
      unsigned base[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int ch = tn * BN + wn * (NI * 32) + j * 32 + fr;
        bb[j] = p.bias[ch];
        base[j] = ch < p.cout_store ? (unsigned)((tm * BM + wm * (MI * 32) + 4 * fh) * p.cout_store + ch) * 4u : W4_HOOB;
      }
      if (!ragged) {
        unsigned row_bs = (unsigned)__builtin_amdgcn_readfirstlane((int)row_b);
        asm volatile("" : "+s"(row_bs));
        float rr[NI][MI][16];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              rr[j][i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, base[j], W4_ROW_OFF(i, r), 0));
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const unsigned keep_n = base[j] != W4_HOOB ? 0x7FFFFFFFu : 0u;
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned o = __float_as_uint(fmaxf(fmaf(acc[i][j][r], tot_unscale, bb[j] + rr[j][i][r]), floor_v));
              unsigned mk;      // (an asm max: as a plain max the compiler builds one reduction tree and keeps every value alive for it)
              asm volatile("v_and_b32 %0, %2, %3\n\tv_max_u32 %1, %1, %0" : "=&v"(mk), "+v"(out_bits) : "v"(o), "v"(keep_n));
              __builtin_amdgcn_raw_buffer_store_b32(o, o_rsrc, base[j], W4_ROW_OFF(i, r), 0);
            }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const unsigned keep_n = base[j] != W4_HOOB ? 0x7FFFFFFFu : 0u;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            float r1[16];
            unsigned off = base[j] + (unsigned)(i * 32) * row_b;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              r1[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off, 0, 0));
              off += ((r & 3) == 3 ? 5u : 1u) * row_b;
            }
            off = base[j] + (unsigned)(i * 32) * row_b;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned o = __float_as_uint(fmaxf(fmaf(acc[i][j][r], tot_unscale, bb[j] + r1[r]), floor_v));
              const int pix = wm * (MI * 32) + i * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
              out_bits = max(out_bits, o & (tm * BM + pix < M ? keep_n : 0u));
              __builtin_amdgcn_raw_buffer_store_b32(o, o_rsrc, off, 0, 0);
              off += ((r & 3) == 3 ? 5u : 1u) * row_b;
            }
          }
        }
      }
    }
    if ((unsigned)next_tile >= (unsigned)n_tiles) break;
    tile = next_tile;
    w_row = w_row_next;
  }
  if (p.out_max) publish_abs_max(p.out_max, out_bits);
#undef W4_H_SETUP
#undef W4_H_OFF
#undef W4_H_ZERO
#undef W4_P_SETUP
#undef W4_READ_X
#undef W4_LOAD_W
#undef W4_MFMA
#undef W4_PIN
#undef W4_ROW_OFF
#undef W4_STORE_PATCH
#undef W4_SLOT_BODY
#undef W4_SLICE
}

template <int WI, int HI>
hipError_t launch_w4_cfg(const ConvLaunch& c, hipStream_t s) {
  const long M = (long)c.n_img * c.H * c.W;
  const int tiles_m = (int)((M + W4_BM - 1) / W4_BM), tiles_n = c.cout_store / W4_BN;
  const int n_tiles = tiles_m * tiles_n;
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_w4_kernel<WI, HI>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL((conv_w4_kernel<WI, HI>), dim3(grid), dim3(256), W4_LDS, s, c, tiles_n, n_tiles);
  return hipGetLastError();
}

}  // namespace

// (the map sizes are template parameters - taps and padded rows are immediates of the fragment reads: the backbone's 12x12 and 6x6)
bool conv_w4_applicable(const ConvLaunch& c) {
  return !(c.no_resident & 2) && c.w_split && c.split_unscale > 0.f && c.ksize == 3 && c.stride == 1 && c.pad == 1 && c.cslice == 32 &&
         c.cin % 32 == 0 && c.cin >= 64 && c.cout_store % W4_BN == 0 && c.cout_pad >= c.cout_store && !c.out_nchw && c.splits == 0 &&
         ((c.W == 12 && c.H == 12) || (c.W == 6 && c.H == 6)) && c.H == c.Ho && c.W == c.Wo && c.k_pad == 9 * c.cin && c.tile_counter &&
         c.num_cu > 0 && (size_t)c.n_img * c.H * c.W * c.cin * sizeof(float) < 0x7FFFFF00ull &&
         (size_t)c.n_img * c.H * c.W * c.cout_store * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_w4(const ConvLaunch& c, hipStream_t s) {
  if (!conv_w4_applicable(c)) return hipErrorInvalidValue;
  return c.W == 12 ? launch_w4_cfg<12, 12>(c, s) : launch_w4_cfg<6, 6>(c, s);
}

}  // namespace ut
