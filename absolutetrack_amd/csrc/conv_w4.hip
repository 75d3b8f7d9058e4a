// 3x3 / stride-1 convolutions from 64 input channels up to a multiple of 128 output channels (layer3's nine 128 -> 128 and
// layer4's three 256 -> 256 convolutions, lib/models/backbone_resnet.py:56-72 at 12x12x128 and 6x6x256: 44 % of a step) in the
// split-fp16 arithmetic of conv_split.hip, as FOUR waves - one per SIMD, each with the SIMD's whole 512-entry register file - and
// NO synchronisation inside a 32-channel slice.
//
// conv_split_kernel<256, 128, 4, 2, true> (eight waves of 64 x 64) pays ~1,600 cycles per 32-deep chunk on top of its MFMAs
// whatever their number: every chunk's weights go global -> LDS and are published to all eight waves through a counter, and a
// wave's 12 MFMAs per k-step have 8 fragment reads to wait for.  Here
//   * a wave owns 128 pixels x 64 output channels of the 256 x 128 tile: 24 MFMAs per k-step on 8 pixel-fragment reads
//     (half the LDS reads per MFMA);
//   * the weights never touch LDS: a wave loads the fragments of ITS 64 output channels straight from the fragment-ordered
//     planes into registers (one coalesced 1-KB load per fragment, two k-steps ahead; the two waves of a channel half hit in
//     L1 behind each other) - no weight ring, no counter, nothing to publish;
//   * the input patch of the next slice (320 rows x 32 channels) comes global -> registers -> LDS: every thread loads ten
//     float4, splits them into the two fp16 pieces and stores the pieces where the fragment reads expect them - one load, one
//     conversion and two 8-byte LDS stores per k-step, between the MFMAs; no LDS-DMA, no in-place conversion pass;
//   * ONE barrier per slice (every 432 MFMAs of a wave): behind it the patch just written is read, the patch just read is
//     written again.
// Every vector-memory operation is a compiler-visible load, so the waits are counted by the compiler; the order of a k-step's
// instructions is pinned slot by slot (one MFMA per slot).
// Same tensors, same fragment layouts and - per output element - the same products in the same order as
// conv_split_kernel<256, 128, 4, 2, true>: bit-identical results.
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
constexpr int W4_HROWS = 320;                  // patch rows per slice (256 + 2 * (image width + 1) <= 320: width <= 31)
constexpr int W4_STAGE = W4_HROWS * 128;       // one slice patch as fp16 piece pairs: 40 KB
constexpr int W4_ZROW = 2 * W4_STAGE;          // 256 bytes of zeros: the target of out-of-image taps, on the reader's own banks
constexpr int W4_SLOT = W4_ZROW + 256;         // the next tile's index
constexpr int W4_LDS = W4_SLOT + 16;
constexpr int W4_NLOAD = W4_HROWS * 8 / 256;   // float4 patch loads per thread and slice: 10
constexpr int W4_NSTG = 5;                     // patch loads in flight per thread (loaded in k-step q, split in k-step q + 4)
constexpr unsigned W4_HOOB = 0x80000000u;      // out-of-range offset that stays out of range with a slice offset added
static_assert(W4_ZROW % 256 == 0, "zero block bank-row aligned");

__device__ __forceinline__ int w4_fdiv(int n, int d, float inv_d) {
  int q = (int)((float)n * inv_d);
  int r = n - q * d;
  if (r < 0) --q;
  if (r >= d) ++q;
  return q;
}
// the pieces of a * s and b * s for a power of two s (conv_split.hip::split_pair_scaled)
__device__ __forceinline__ void w4_split_pair(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2w h = __builtin_bit_cast(f16x2w, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

__global__ __launch_bounds__(256, 1) void conv_w4_kernel(ConvLaunch p, int tiles_n, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MI = W4_MI, NI = W4_NI, BM = W4_BM, BN = W4_BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;

  const int M = p.n_img * p.Ho * p.Wo;
  const int hw = p.Ho * p.Wo;
  const int wimg = p.W;
  const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)wimg;
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

  const int grid = gridDim.x;
  int slot = blockIdx.x;
  if ((grid & 7) == 0) slot = (blockIdx.x & 7) * (grid >> 3) + (blockIdx.x >> 3);      // first round XCD-contiguous

  // ---- patch stream: thread t loads group t & 7 of rows 32 j + (t >> 3), j = 0 .. 9: one per-lane byte offset for row (t >> 3) of the
  // tile being fetched (slice 0) plus j x 32 rows.  Rows in front of the tensor (a negative pixel: the offset wraps far beyond
  // 2^31) and behind it are beyond the descriptor's range: they load zeros, with no compare per row.
  unsigned h_base = 0;
  const unsigned h_step = (unsigned)(32 * p.cin) * 4u;
  const unsigned h_wpos = (unsigned)((tid >> 3) * 128 + ((((tid & 7) >> 1) ^ ((tid >> 4) & 7)) << 4) + (tid & 1) * 8);
#define W4_H_SETUP(TILE) { h_base = (unsigned)(((((TILE) / tiles_n) * BM - wimg - 1) + (tid >> 3)) * p.cin + 4 * (tid & 7)) * 4u; }
#define W4_H_OFF(J) (h_base + (unsigned)(J) * h_step)
  // ---- read side
  unsigned rmask[MI];             // 9 validity bits (tap order) of the lane's pixel in fragment i of the tile being computed
  int lrow[MI];                   // its patch row for the centre tap
#pragma unroll
  for (int i = 0; i < MI; ++i) { lrow[i] = wm * (MI * 32) + i * 32 + fr + wimg + 1; rmask[i] = 0; }
#define W4_MASK(TILE)                                                                                \
  {                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                 \
      const int m_ = ((TILE) / tiles_n) * BM + wm * (MI * 32) + i * 32 + fr;                         \
      const bool in_ = m_ < M && (unsigned)(TILE) < (unsigned)n_tiles;                               \
      const int mm_ = in_ ? m_ : 0;                                                                  \
      const int img_ = w4_fdiv(mm_, hw, inv_hw);                                                     \
      const int rem_ = mm_ - img_ * hw;                                                              \
      const int y_ = w4_fdiv(rem_, wimg, inv_wo), x_ = rem_ - y_ * wimg;                             \
      unsigned mk_ = 0;                                                                              \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                \
        const bool ok_ = (unsigned)(y_ + t / 3 - 1) < (unsigned)p.H && (unsigned)(x_ + t % 3 - 1) < (unsigned)wimg; \
        mk_ |= (ok_ ? 1u : 0u) << t;                                                                 \
      }                                                                                              \
      rmask[i] = in_ ? mk_ : 0u;                                                                     \
    }                                                                                                \
  }

  f32x16w acc[MI][NI];
  u32x4w xp[2][MI][2];            // pixel fragments (first piece, remainder), two k-steps
  u32x4w wf[3][NI][2];            // weight fragments (plane 0, plane 1), three k-steps
  float4 stg[W4_NSTG];            // patch values between their load and their split
  const unsigned w_lane = (unsigned)lane * 16u;

  // pixel fragments of block I for k-step half S of tap TAP out of patch buffer RB into set SET
#define W4_READ_X(SET, I, S, TAP, RB)                                                                \
  {                                                                                                  \
    const int dy_ = ((TAP) * 11) >> 5, dx_ = (TAP) - 3 * dy_;                                        \
    int sh_ = (dy_ - 1) * wimg + dx_ - 1;                                                            \
    /* (opaque here: the address arithmetic below then stays in this slot; left to itself the compiler computes the fragment  \
       addresses of all nine taps at the top of the slice and carries 144 of them in registers) */   \
    asm volatile("" : "+s"(sh_));                                                                    \
    const int row_ = lrow[I] + sh_;                                                                  \
    const unsigned a_ = (RB) + (unsigned)(row_ * 128) + (unsigned)((((2 * (S) + fh) ^ ((row_ >> 1) & 7))) << 4); \
    const unsigned a0_ = ((rmask[I] >> (TAP)) & 1u) ? a_ : (unsigned)W4_ZROW + (a_ & 255u);          \
    xp[SET][I][0] = *reinterpret_cast<const u32x4w*>(smem + a0_);                                    \
    xp[SET][I][1] = *reinterpret_cast<const u32x4w*>(smem + (a0_ ^ 64u));                            \
  }
  // weight fragment IDX (block IDX / 2, plane IDX % 2) of chunk CH, k-step half S, of the tile column at byte offset WROW
#define W4_LOAD_W(SET, IDX, CH, S, WROW)                                                             \
  {                                                                                                  \
    const unsigned so_ = (WROW) + (unsigned)(((wn * NI + (IDX) / 2) * n_chunks + (CH)) * 4096 + ((S) * 2 + (IDX) % 2) * 1024); \
    wf[SET][(IDX) / 2][(IDX) % 2] = __builtin_bit_cast(u32x4w, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, so_, 0)); \
  }
#define W4_MFMA(XSET, WSET, N)                                                                       \
  {                                                                                                  \
    constexpr int pr_ = (N) / (MI * NI), ij_ = (N) % (MI * NI), i_ = ij_ / NI, j_ = ij_ % NI;        \
    constexpr int wp_ = pr_ == 1 ? 1 : 0, xq_ = pr_ == 0 ? 1 : 0;      /* small terms first: x1 w0, x0 w1, x0 w0 */ \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, xp[XSET][i_][xq_]),             \
                                                         __builtin_bit_cast(f16x8w, wf[WSET][j_][wp_]), acc[i_][j_], 0, 0, 0); \
  }
#define W4_PIN() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

  int tile = slot;
  int next_tile = 0;
  int cur_buf = 0;                // patch buffer (0 / 1) of the slice being computed
  unsigned out_bits = 0;
  const unsigned slot_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem + (unsigned)W4_SLOT;

  // ---- prologue: the first tile's first patch (exposed once per workgroup), the weights of its first two k-steps and the pixel
  // fragments of its first
  if (tid < 16) *reinterpret_cast<u32x4w*>(smem + W4_ZROW + tid * 16) = u32x4w{0, 0, 0, 0};
  W4_H_SETUP(tile);
  W4_MASK(tile);
#pragma unroll
  for (int j = 0; j < W4_NLOAD; ++j) {
    const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, W4_H_OFF(j), 0, 0));
    unsigned a0, b0, a1, b1;
    w4_split_pair(v.x, v.y, x_scale, a0, b0);
    w4_split_pair(v.z, v.w, x_scale, a1, b1);
    *reinterpret_cast<u32x2w*>(smem + j * 4096 + h_wpos) = u32x2w{a0, a1};
    *reinterpret_cast<u32x2w*>(smem + ((j * 4096 + h_wpos) ^ 64u)) = u32x2w{b0, b1};
  }
  unsigned w_row = (unsigned)((tile % tiles_n) * (BN / 32)) * (unsigned)n_chunks * 4096u;     // byte offset of the tile column's planes
  unsigned w_row_next = w_row;
#pragma unroll
  for (int idx = 0; idx < 4; ++idx) { W4_LOAD_W(0, idx, 0, 0, w_row) }
#pragma unroll
  for (int idx = 0; idx < 4; ++idx) { W4_LOAD_W(1, idx, 0, 1, w_row) }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) { W4_READ_X(0, i, 0, 0, 0u) }

  // The epilogue of a tile runs UNDER the first slice of the workgroup's next tile: the finished accumulators move to a second
  // register set (`prev`), and that slice's k-steps carry, in their free slots, sixteen units of (8 residual loads, two k-steps
  // later 8 x scale + bias + residual, ReLU, store).  With one wave per SIMD nothing else would overlap it, and every workgroup of
  // a launch finishes its tiles at the same time: as a phase of its own the epilogue's 256 KB per CU met the other 255 CUs' at
  // the memory (a third of a tile's time).  The last tile of a workgroup, and a tile that reaches beyond the tensor, take the
  // stand-alone epilogue behind the loop body.
  f32x16w prev[MI][NI];
  float rq[3][8];                 // residual values of the units in flight
  unsigned prev_base[NI] = {0, 0};
  float prev_bb[NI] = {0.f, 0.f};
  unsigned prev_keep[NI] = {0, 0};
  bool has_prev = false;
  const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();
  const unsigned row_b = (unsigned)p.cout_store * 4u;
#define W4_ROW_OFF(I, R) ((unsigned)((I) * 32 + 8 * ((R) >> 2) + ((R) & 3)) * row_bs)

  // one slice: 18 k-steps of 24 slots, one MFMA per slot; EPI: the previous tile's epilogue rides along
#define W4_SLICE(EPI)                                                                                \
      _Pragma("clang loop unroll(full)") for (int q = 0; q < 18; ++q) {                                               \
        const int xs = q & 1, ws = q % 3;                                                            \
        /* (q + 1): the k-step whose pixel fragments are read now; (q + 2): the k-step whose weights are loaded now */ \
        const int q1 = q + 1, q2 = q + 2;                                                            \
        if (q == 17) {                                                                               \
          /* every wave has written its part of the next patch (k-steps 4 .. 13) and has read its last fragments of this one \
             (in k-step 16): ONE barrier per slice, in front of the first reads of the next patch */ \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
          __builtin_amdgcn_s_barrier();                                                              \
          asm volatile("" ::: "memory");                                                             \
          if (sl == 0) {                                                                             \
            int nv;                                                                                  \
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nv) : "v"(slot_addr) : "memory"); \
            next_tile = __builtin_amdgcn_readfirstlane(nv);                                          \
            w_row_next = (unsigned)((next_tile % tiles_n) * (BN / 32)) * (unsigned)n_chunks * 4096u; \
          }                                                                                          \
          if (last_slice) W4_MASK(next_tile);       /* the fragments read in this k-step are the next tile's */ \
        }                                                                                            \
        W4_SLOT_BODY(EPI, 0) W4_SLOT_BODY(EPI, 1) W4_SLOT_BODY(EPI, 2) W4_SLOT_BODY(EPI, 3) W4_SLOT_BODY(EPI, 4) W4_SLOT_BODY(EPI, 5) \
        W4_SLOT_BODY(EPI, 6) W4_SLOT_BODY(EPI, 7) W4_SLOT_BODY(EPI, 8) W4_SLOT_BODY(EPI, 9) W4_SLOT_BODY(EPI, 10) W4_SLOT_BODY(EPI, 11) \
        W4_SLOT_BODY(EPI, 12) W4_SLOT_BODY(EPI, 13) W4_SLOT_BODY(EPI, 14) W4_SLOT_BODY(EPI, 15) W4_SLOT_BODY(EPI, 16) W4_SLOT_BODY(EPI, 17) \
        W4_SLOT_BODY(EPI, 18) W4_SLOT_BODY(EPI, 19) W4_SLOT_BODY(EPI, 20) W4_SLOT_BODY(EPI, 21) W4_SLOT_BODY(EPI, 22) W4_SLOT_BODY(EPI, 23) \
      }
  // Slot N of k-step q.  Slots 0 .. 7: the next k-step's pixel fragments (a block per two slots); 8 .. 11: the weights of the k-step
  // after next (+ two residual loads each of epilogue unit q); 12: a patch load; 14: the split and store of the patch load of four
  // k-steps ago; 15 .. 22: one value each of epilogue unit q - 2.  Every load of a k-step is issued before its stores.
#define W4_SLOT_BODY(EPI, N)                                                                         \
        {                                                                                            \
          if ((N) < 8 && ((N) & 1) == 0 && q1 < 18) { W4_READ_X(q1 & 1, (N) / 2, q1 & 1, q1 >> 1, rbuf) } \
          if ((N) < 8 && ((N) & 1) == 0 && q1 == 18) { W4_READ_X(0, (N) / 2, 0, 0, wbuf) }           \
          if ((N) >= 8 && (N) < 12 && q2 < 18) { W4_LOAD_W(q2 % 3, (N) - 8, ch0 + (q2 >> 1), q2 & 1, w_row) } \
          if ((N) >= 8 && (N) < 12 && q2 >= 18) { W4_LOAD_W(q2 % 3, (N) - 8, ch_after, q2 & 1, row_after) } \
          if ((EPI) && (N) >= 8 && (N) < 12 && q < 16) {                                             \
            constexpr int dummy_ = 0; (void)dummy_;                                                  \
            const int uj_ = q >> 3, ui_ = (q >> 1) & 3, uh_ = q & 1;                                 \
            _Pragma("unroll") for (int e2 = 0; e2 < 2; ++e2) {                                       \
              const int e_ = ((N) - 8) * 2 + e2, r_ = 8 * uh_ + e_;                                  \
              rq[q % 3][e_] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, prev_base[uj_], W4_ROW_OFF(ui_, r_), 0)); \
            }                                                                                        \
          }                                                                                          \
          if ((N) == 12 && q < W4_NLOAD)                                                             \
            stg[q % W4_NSTG] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, W4_H_OFF(q), f_soff, 0)); \
          if ((N) == 13 && q == 10 && sl == 0 && tid == 0) *reinterpret_cast<int*>(smem + W4_SLOT) = grid + ticket; \
          if ((N) == 14 && q >= 4 && q < 4 + W4_NLOAD) {                                             \
            const float4 v_ = stg[(q - 4) % W4_NSTG];                                                \
            unsigned a0_, b0_, a1_, b1_;                                                             \
            w4_split_pair(v_.x, v_.y, x_scale, a0_, b0_);                                            \
            w4_split_pair(v_.z, v_.w, x_scale, a1_, b1_);                                            \
            const unsigned wa_ = wbuf + (unsigned)((q - 4) * 4096) + h_wpos;                         \
            *reinterpret_cast<u32x2w*>(smem + wa_) = u32x2w{a0_, a1_};                               \
            *reinterpret_cast<u32x2w*>(smem + (wa_ ^ 64u)) = u32x2w{b0_, b1_};                       \
          }                                                                                          \
          if ((EPI) && (N) >= 15 && (N) < 23 && q >= 2) {                                            \
            const int u_ = q - 2, uj_ = u_ >> 3, ui_ = (u_ >> 1) & 3, uh_ = u_ & 1;                  \
            const int e_ = (N) - 15, r_ = 8 * uh_ + e_;                                              \
            const unsigned o_ = __float_as_uint(fmaxf(fmaf(prev[ui_][uj_][r_], tot_unscale, prev_bb[uj_] + rq[u_ % 3][e_]), floor_v)); \
            /* (an asm max: as a plain max the compiler rebuilds the 128 of a tile as one reduction tree behind the slice and keeps \
               every stored value in a register until then) */                                       \
            unsigned mk_;                                                                            \
            asm volatile("v_and_b32 %0, %2, %3\n\tv_max_u32 %1, %1, %0" : "=&v"(mk_), "+v"(out_bits) : "v"(o_), "v"(prev_keep[uj_])); \
            __builtin_amdgcn_raw_buffer_store_b32(o_, o_rsrc, prev_base[uj_], W4_ROW_OFF(ui_, r_), 0); \
          }                                                                                          \
          W4_PIN();                                                                                  \
          W4_MFMA(xs, ws, N);                                                                        \
          W4_PIN();                                                                                  \
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
      // the patch fetched during this slice: the next slice of this tile, or slice 0 of the next tile
      if (last_slice) W4_H_SETUP(next_tile);
      const unsigned f_soff = last_slice ? 0u : (unsigned)(sl + 1) * 128u;
      unsigned rbuf = (unsigned)__builtin_amdgcn_readfirstlane(cur_buf * W4_STAGE), wbuf = (unsigned)__builtin_amdgcn_readfirstlane((cur_buf ^ 1) * W4_STAGE);
      // (opaque per slice: the compiler otherwise keeps every tap's fragment addresses of both buffers, and every row offset of
      // the epilogue, in registers across the whole kernel - a hundred registers the k-steps need)
      asm volatile("" : "+s"(rbuf), "+s"(wbuf));
      unsigned row_bs = (unsigned)__builtin_amdgcn_readfirstlane((int)row_b);
      asm volatile("" : "+s"(row_bs));
      const int ch0 = sl * 9;
      // the weight fragments of the two k-steps behind this slice: the next slice's first chunk, or the next tile's
      const int ch_after = last_slice ? 0 : ch0 + 9;
      const unsigned row_after = last_slice ? w_row_next : w_row;
      if (sl == 0 && has_prev) {
        W4_SLICE(1)
      } else {
        W4_SLICE(0)
      }
      cur_buf ^= 1;
    }
    // ---- the tile's accumulators: to `prev` (their epilogue rides under the next tile's first slice), or the stand-alone epilogue
    {
      const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
      const bool ragged = tm * BM + BM > M;
      const bool more = (unsigned)next_tile < (unsigned)n_tiles;
      float bb[NI];
      unsigned base[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int ch = tn * BN + wn * (NI * 32) + j * 32 + fr;
        bb[j] = p.bias[ch];
        base[j] = ch < p.cout_store ? (unsigned)((tm * BM + wm * (MI * 32) + 4 * fh) * p.cout_store + ch) * 4u : W4_HOOB;
      }
      if (more && !ragged) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) prev[i][j] = acc[i][j];
#pragma unroll
        for (int j = 0; j < NI; ++j) { prev_base[j] = base[j]; prev_bb[j] = bb[j]; prev_keep[j] = base[j] != W4_HOOB ? 0x7FFFFFFFu : 0u; }
        has_prev = true;
      } else {
        has_prev = false;
        // pixels / channels beyond the tensor: per-access offsets (the scalar offset is not part of the descriptor's range check)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const unsigned keep_n = base[j] != W4_HOOB ? 0x7FFFFFFFu : 0u;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            float rr[16];
            unsigned off = base[j] + (unsigned)(i * 32) * row_b;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              rr[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off, 0, 0));
              off += ((r & 3) == 3 ? 5u : 1u) * row_b;
            }
            off = base[j] + (unsigned)(i * 32) * row_b;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned o = __float_as_uint(fmaxf(fmaf(acc[i][j][r], tot_unscale, bb[j] + rr[r]), floor_v));
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
#undef W4_SLICE
#undef W4_SLOT_BODY
#undef W4_ROW_OFF
  if (p.out_max) publish_abs_max(p.out_max, out_bits);
#undef W4_H_SETUP
#undef W4_H_OFF
#undef W4_MASK
#undef W4_READ_X
#undef W4_LOAD_W
#undef W4_MFMA
#undef W4_PIN
}

}  // namespace

bool conv_w4_applicable(const ConvLaunch& c) {
  return !(c.no_resident & 2) && c.w_split && c.split_unscale > 0.f && c.ksize == 3 && c.stride == 1 && c.pad == 1 && c.cslice == 32 &&
         c.cin % 32 == 0 && c.cin >= 64 && c.cout_store % W4_BN == 0 && c.cout_pad >= c.cout_store && !c.out_nchw && c.splits == 0 &&
         c.W <= 31 && c.H == c.Ho && c.W == c.Wo && c.k_pad == 9 * c.cin && c.tile_counter && c.num_cu > 0 &&
         (size_t)c.n_img * c.H * c.W * c.cin * sizeof(float) < 0x7FFFFF00ull &&
         (size_t)c.n_img * c.H * c.W * c.cout_store * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_w4(const ConvLaunch& c, hipStream_t s) {
  if (!conv_w4_applicable(c)) return hipErrorInvalidValue;
  const long M = (long)c.n_img * c.H * c.W;
  const int tiles_m = (int)((M + W4_BM - 1) / W4_BM), tiles_n = c.cout_store / W4_BN;
  const int n_tiles = tiles_m * tiles_n;
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(conv_w4_kernel, dim3(grid), dim3(256), W4_LDS, s, c, tiles_n, n_tiles);
  return hipGetLastError();
}

}  // namespace ut
