// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces every Conv2d(+BatchNorm2d)(+residual)(+ReLU) of the reference after the stem:
// lib/models/backbone_resnet.py:56-72 (BasicBlock), lib/models/model_utils.py:134 (projection),
// :141-163 (fusion), lib/models/temporal.py:31-38, lib/models/model_utils.py:195-208 (regressor).
//
// GEMM view: M = n_img*Ho*Wo output pixels, N = cout, K = taps*cin, k ordered (channel slice, tap,
// channel) - see ut_kernels.h.  Activations are NHWC so a k-run of 4 channels is one 16-byte load;
// weights are pre-packed [cout_pad][k_pad] (k contiguous) with BatchNorm folded in.
//
// A workgroup (4 waves, 256 threads) computes BM x BN output tiles and walks K in chunks of 32.  Operands go
// global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction): the im2col gather for the
// pixels, plain rows for the weights; out-of-image taps and rows beyond M get an out-of-range buffer offset and
// arrive as zeros, so the load path has no branches.  LDS rows are 128 B, unpadded, their 16-byte chunks
// XOR-swizzled by (row >> 1) & 7 on the SOURCE side (which chunk a lane fetches), which makes the ds_read_b128
// fragment reads conflict free under gfx950's 16-lane read groups.  Two stages, one barrier per chunk.
// Lane l of a wave holds row (l&31) of the fragment and k-half (l>>5); the 4 floats of a b128 read feed
// 4 consecutive MFMAs (the k order inside the 8-run is permuted identically for A and B).
//
// Workgroups are PERSISTENT: a grid of (CUs x resident workgroups) takes tiles from a device-wide queue (first
// round: static XCD-contiguous slots; afterwards one atomic ticket per workgroup per tile, requested a tile
// ahead by wave 0 and handed to the other waves through an LDS word), and the first chunk, bias and residual of
// the next tile are fetched under the last chunk of the current one.  A 64-cycle MFMA makes operand traffic
// cheap; what costs throughput is every cycle the matrix pipe waits for a tile prologue (index math,
// first-touch latency, residual fetch) or for a store-bound epilogue.
// Accumulators start at bias (+ residual); epilogue = (ReLU) + store.  The weights are the MFMA "A" operand
// and the pixels the "B" operand, so in the C layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
// a lane owns one pixel and register quads are 4 consecutive channels: 16-byte NHWC loads and stores.
// When the channel slice of the k order equals the chunk width (cin % 32 == 0: every backbone convolution) the
// C32 instantiation folds the thread's 16-byte k-group into the row offsets, which makes the (slice, tap)
// bookkeeping of the chunk loop wave-uniform: it runs on the scalar unit and the loop body has no branch.
// Tile shapes: 128x128 (cout > 64), 128x64 (cout <= 64, three workgroups per CU), 64x128 for launches with few
// tiles (projection and head: 74 k pixels); layer1 (3x3, 32 -> 32) runs in conv_patch.hip instead.
#include <atomic>

#include "ut_kernels.h"

namespace ut {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;

constexpr int BK = 32;
// LDS row stride in floats.  LDS-DMA staging (buffer_load ... lds) writes 64 lanes x 16 B = 8 whole rows
// linearly, so rows are unpadded (32 floats) and the 16-byte chunks of a row are XOR-swizzled:
// chunk c of row r lives at position c ^ ((r >> 1) & 7).  The permutation is applied on the SOURCE side
// (which global chunk a lane fetches); 16-lane read groups then hit 16 distinct 16-byte bank slots.
constexpr int LDS_ROW = BK;

// One LDS-DMA piece: 64 lanes x 16 bytes from a buffer (per-lane byte offset, out-of-range -> zeros) straight
// into LDS at lds_addr + lane*16.  Inline asm on purpose: with the builtin hipcc treats the pending LDS write
// as aliasing every ds_read and drains vmcnt(0) in front of the fragment reads of the CURRENT buffer, which
// serialises the whole prefetch.  M0 (the LDS base of the transfer) is written in the statement that uses it
// and restored; the transfer is invisible to the compiler's wait counting, so dma_wait_all() precedes the
// barrier that publishes the buffer.
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, unsigned voffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc)
      : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ u32x4 make_rsrc_words(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);   // stride 0
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}

// n / d for 0 <= n < 2^24 via a float reciprocal and one correction step (exact: |error| <= 1 before it)
__device__ __forceinline__ int fast_div(int n, int d, float inv_d) {
  int q = (int)((float)n * inv_d);
  int r = n - q * d;
  if (r < 0) --q;
  if (r >= d) ++q;
  return q;
}

// SPLITK (latency mode, launches of a few crops): the tile queue holds p.splits entries per output tile, entry
// (split s, tile t) walks the chunks [s, s+1) * n_chunks of K and stores its partial sums (no bias, residual or ReLU:
// the host passes a zero bias and no residual) to slab s of p.out [splits][M][cout_store]; splitk_finish_kernel adds
// the slabs in a fixed order and applies the epilogue.  Not used by the throughput path: its kernels are the
// SPLITK = false instantiations, unchanged.
template <int BM, int BN, int WR, int WC, bool NCHW = false, bool C32 = false, bool SPLITK = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvLaunch p, int tiles_n, int n_tiles, int stagger) {
  static_assert(WR * WC == 4, "4 waves per workgroup");
  constexpr int MI = BM / WR / 32;   // 32x32 accumulator tiles per wave along M
  constexpr int NI = BN / WC / 32;   // ... along N
  constexpr int AP = BM / 32;        // 16-byte loads per thread per chunk for the A tile
  constexpr int BP = BN / 32;
  constexpr int STAGE = (BM + BN) * LDS_ROW;
  constexpr unsigned OOB = 0xFFFFFF00u;
  static_assert(AP + BP <= 8, "the chunk loop places one LDS-DMA piece per MFMA step of its first two groups");

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WC, wn = wave % WC;
  const int g = tid & 7;        // 16-byte position inside the staged LDS row this thread fills
  const int r0 = tid >> 3;      // first tile row this thread stages (then +32 per pass)
  // which 4-float group of the 32-wide k chunk lands there ((r0 + 32*i) >> 1 & 7 is the same for every pass i)
  const int gk = g ^ ((r0 >> 1) & 7);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // provably uniform: LDS-DMA base, M0
  const int fr = lane & 31;     // fragment row (A/B) == accumulator column
  const int fh = lane >> 5;     // k half (A/B) == accumulator row offset 4*fh

  const int M = p.n_img * p.Ho * p.Wo;
  const int hw = p.Ho * p.Wo;
  const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)p.Wo;
  const int taps = p.ksize * p.ksize;
  const int n_splits = SPLITK ? p.splits : 1;
  const int real_tiles = SPLITK ? n_tiles / n_splits : n_tiles;        // output tiles (n_tiles counts queue entries)
  const int n_chunks = p.k_pad / BK / n_splits;                        // chunks walked per queue entry
  // queue entry -> (split, tile row, tile column); entries past the end get a tile row beyond M (every row then
  // fails the range checks and the requests return zeros, as for SPLITK = false where that follows from the division)
#define UT_DECOMP(TILE)                                                                              \
    const int sp_ = SPLITK ? (TILE) / real_tiles : 0;                                                \
    const int t_ = (TILE) - sp_ * real_tiles;                                                        \
    const int tq_ = t_ / tiles_n;                                                                    \
    const int tm_ = (!SPLITK || (TILE) < n_tiles) ? tq_ : (1 << 20), tn_ = t_ - tq_ * tiles_n;       \
    (void)sp_;
  const unsigned b_row_step = (unsigned)(32 * p.k_pad * 4);

  // with no residual the descriptor is empty and every load returns 0
  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.bias), 0, p.res ? (int)((size_t)M * p.cout_store * sizeof(float)) : 0,
      0x00020000);

  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)((size_t)n_splits * M * p.cout_store * sizeof(float)), 0x00020000);

  // tile-queue ticket: a straight-line buffer atomic (only thread 0 has an in-range offset, the range check
  // drops the others) whose result is awaited where it is used, not where it is issued
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.tile_counter, 0, 4, 0x00020000);
  const unsigned q_off = tid == 0 ? 0u : OOB;
  const u32x4 a_words = make_rsrc_words(p.in, (unsigned)((size_t)p.n_img * p.H * p.W * p.cin * sizeof(float)));
  const u32x4 b_words = make_rsrc_words(p.w, (unsigned)((size_t)p.cout_pad * p.k_pad * sizeof(float)));
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_f32*)smem;   // LDS byte address of the staging area

  // XCD-aware tile order: workgroups b, b+8, ... share an XCD (and its L2); give each XCD a contiguous
  // run of tiles per round so that neighbouring tiles (shared halo rows, same weights) meet in one L2.
  const int grid = gridDim.x;
  int slot = blockIdx.x;
  if ((grid & 7) == 0) slot = (blockIdx.x & 7) * (grid >> 3) + (blockIdx.x >> 3);

  // ---- fetch-side state: im2col rows of the tile being FETCHED and its position in K
  int a_pix[AP], a_iy[AP], a_ix[AP];
  unsigned b_off;
  int tap, ch, ch_base;
  // bias and residual of the tile being fetched: requested a tile ahead, combined only when that tile starts
  // (arithmetic at request time would make the compiler wait for the loads in front of the MFMAs)
  u32x4 res_raw[MI][NI][4];
  float4 bias_raw[NI][4];

#define UT_SETUP(TILE)                                                                               \
  {                                                                                                  \
    UT_DECOMP(TILE)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                                 \
      const int m = tm_ * BM + r0 + 32 * i;                                                          \
      const bool ok = m < M;                                                                         \
      const int mm = ok ? m : 0;                                                                     \
      const int img = fast_div(mm, hw, inv_hw);                                                      \
      const int rem = mm - img * hw;                                                                 \
      const int oy = fast_div(rem, p.Wo, inv_wo), ox = rem - oy * p.Wo;                              \
      a_iy[i] = ok ? oy * p.stride - p.pad : -100000; /* rows beyond M never pass the bounds test */ \
      a_ix[i] = ox * p.stride - p.pad;                                                               \
      a_pix[i] = ((img * p.H + a_iy[i]) * p.W + a_ix[i]) * p.cin + (C32 ? 4 * gk : 0);               \
    }                                                                                                \
    b_off = (unsigned)(((tn_ * BN + r0) * p.k_pad + 4 * gk) * 4);                                    \
    /* (slice, tap, channel in slice) of this thread's 4-float group: cslice >= 32 */                \
    /* C32 (channel slice == chunk width): the thread's 4-float group is folded into a_pix, so (tap, ch,    \
       ch_base) advance identically in every lane and the compiler keeps them - and the tap offset - scalar */ \
    tap = 0; ch = C32 ? 0 : 4 * gk; ch_base = 0;                                                     \
    if constexpr (SPLITK) {      /* start at chunk sp_ * n_chunks of K */                            \
      const int k0 = sp_ * n_chunks * BK, per = taps * p.cslice;                                     \
      const int sl = k0 / per, rem = k0 - sl * per;                                                  \
      tap = rem / p.cslice;                                                                          \
      ch += rem - tap * p.cslice;      /* + the thread's own 4-float group when !C32: may cross into the next tap */ \
      ch_base = sl * p.cslice;                                                                       \
      if (ch >= p.cslice) { ch -= p.cslice; ++tap; }                                                 \
      if (tap >= taps) { tap -= taps; ch_base += p.cslice; }                                         \
      b_off += (unsigned)(k0 * 4);                                                                   \
    }                                                                                                \
  }

  // Prologue only: all AP+BP pieces of a chunk in one burst (the steady state places them one per MFMA step).
#define UT_FETCH(DSTBUF)                                                                             \
  {                                                                                                  \
    int dy = 0, dx = 0;                                                                              \
    if (p.ksize == 3) { dy = (tap * 11) >> 5; dx = tap - 3 * dy; } /* tap/3 for tap < 32 */         \
    const int tap_off = (dy * p.W + dx) * p.cin + ch_base + ch;                                      \
    const unsigned dst_ = smem_addr + (unsigned)(((DSTBUF) * STAGE + 8 * wave_u * LDS_ROW) * 4);     \
    _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                                 \
      const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;                                                \
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;                  \
      const unsigned off = ok ? (unsigned)(a_pix[i] + tap_off) * 4u : OOB;                           \
      dma16(a_words, dst_ + 32 * i * LDS_ROW * 4, off);                                              \
    }                                                                                                \
    _Pragma("unroll") for (int i = 0; i < BP; ++i)                                                   \
      dma16(b_words, dst_ + (BM + 32 * i) * LDS_ROW * 4, b_off + i * b_row_step);                    \
    UT_ADVANCE();                                                                                    \
  }
  /* (slice, tap, channel) of the next chunk */
#define UT_ADVANCE()                                                                                 \
  {                                                                                                  \
    b_off += BK * 4;                                                                                 \
    ch += BK;                                                                                        \
    if (ch >= p.cslice) { ch -= p.cslice; ++tap; }                                                   \
    if (tap >= taps) { tap -= taps; ch_base += p.cslice; }                                           \
  }

  // One LDS-DMA piece of the NEXT chunk per MFMA step, in two halves - the per-lane offset (vector ALU) and the
  // transfer itself - so that each half can sit in its own MFMA gap.  The per-chunk tap arithmetic is done with
  // piece 0 and kept in f_tap_off / f_dst.
  int f_tap_off = 0, f_dy = 0, f_dx = 0;
  unsigned f_dst = 0, f_off = 0;
  // the same piece in two halves: the per-lane offset (vector ALU) and the transfer itself, so that each half can
  // sit in its own MFMA gap
#define UT_PIECE_ADDR(IDX, DSTBUF)                                                                   \
  {                                                                                                  \
    if ((IDX) == 0) {                                                                                \
      f_dy = 0; f_dx = 0;                                                                            \
      if (p.ksize == 3) { f_dy = (tap * 11) >> 5; f_dx = tap - 3 * f_dy; }                           \
      f_tap_off = (f_dy * p.W + f_dx) * p.cin + ch_base + ch;                                        \
      f_dst = smem_addr + (unsigned)(((DSTBUF) * STAGE + 8 * wave_u * LDS_ROW) * 4);                 \
    }                                                                                                \
    if constexpr ((IDX) < AP) {                                                                      \
      constexpr int i = (IDX);                                                                       \
      const int iy = a_iy[i] + f_dy, ix = a_ix[i] + f_dx;                                            \
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;                  \
      f_off = ok ? (unsigned)(a_pix[i] + f_tap_off) * 4u : OOB;                                      \
    } else if constexpr ((IDX) < AP + BP) {                                                          \
      f_off = b_off + ((IDX) - AP) * b_row_step;                                                     \
    }                                                                                                \
  }
#define UT_PIECE_ISSUE(IDX)                                                                          \
  {                                                                                                  \
    if constexpr ((IDX) < AP) {                                                                      \
      dma16(a_words, f_dst + 32 * (IDX) * LDS_ROW * 4, f_off);                            \
    } else if constexpr ((IDX) < AP + BP) {                                                          \
      dma16(b_words, f_dst + (BM + 32 * ((IDX) - AP)) * LDS_ROW * 4, f_off);              \
    }                                                                                                \
    if ((IDX) == AP + BP - 1) UT_ADVANCE();                                                          \
  }

  // MFMA C layout with the operands as above: lane = pixel (column fr of the 32-pixel fragment), register e =
  // output channel (e&3) + 8*(e>>2) + 4*fh of the 32-channel fragment.  Four consecutive registers are four
  // consecutive channels of one pixel: one 16-byte access in NHWC.
#define UT_INIT_LOAD(TILE)                                                                           \
  {                                                                                                  \
    UT_DECOMP(TILE)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
      _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4)                                               \
        bias_raw[j][g4] = *reinterpret_cast<const float4*>(                                          \
            p.bias + tn_ * BN + wn * (NI * 32) + j * 32 + 8 * g4 + 4 * fh); /* padded to cout_pad */ \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                 \
      const int m = tm_ * BM + wm * (MI * 32) + i * 32 + fr;                                         \
      const bool m_ok = m < M;                                                                       \
      _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                               \
        _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                           \
          const int n = tn_ * BN + wn * (NI * 32) + j * 32 + 8 * g4 + 4 * fh;                        \
          const unsigned off = (m_ok && n < p.cout_store) ? (unsigned)(m * p.cout_store + n) * 4u : OOB; \
          res_raw[i][j][g4] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0);              \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
  }
  // the same requests for one accumulator (i, j) = (Q / NI, Q % NI) only: slipped between the MFMA steps of a
  // tile's last chunk.  Safe for TILE >= n_tiles (no next tile): rows are beyond M, so every offset is out of range.
#define UT_INIT_LOAD_PART(TILE, Q)                                                                   \
  if constexpr ((Q) < MI * NI) {                                                                     \
    constexpr int i = (Q) / NI, j = (Q) % NI;                                                        \
    UT_DECOMP(TILE)                                                                                  \
    if constexpr (i == 0) {                                                                          \
      _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4)                                               \
        bias_raw[j][g4] = *reinterpret_cast<const float4*>(                                          \
            p.bias + tn_ * BN + wn * (NI * 32) + j * 32 + 8 * g4 + 4 * fh);                          \
    }                                                                                                \
    const int m = tm_ * BM + wm * (MI * 32) + i * 32 + fr;                                           \
    const bool m_ok = m < M;                                                                         \
    _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                               \
      const int n = tn_ * BN + wn * (NI * 32) + j * 32 + 8 * g4 + 4 * fh;                            \
      const unsigned off = (m_ok && n < p.cout_store) ? (unsigned)(m * p.cout_store + n) * 4u : OOB; \
      res_raw[i][j][g4] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0);                  \
    }                                                                                                \
  }
  /* NB: __builtin_bit_cast on a vector ELEMENT (r.y) miscompiles to a splat of r.x (ROCm 7.2): __uint_as_float */
#define UT_INIT_COMBINE()                                                                            \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                     \
    _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
      _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                             \
        acc[i][j][4 * g4 + 0] = bias_raw[j][g4].x + __uint_as_float(res_raw[i][j][g4].x);            \
        acc[i][j][4 * g4 + 1] = bias_raw[j][g4].y + __uint_as_float(res_raw[i][j][g4].y);            \
        acc[i][j][4 * g4 + 2] = bias_raw[j][g4].z + __uint_as_float(res_raw[i][j][g4].z);            \
        acc[i][j][4 * g4 + 3] = bias_raw[j][g4].w + __uint_as_float(res_raw[i][j][g4].w);            \
      }

  /* every piece of the next chunk has landed before the barrier publishes it */
#define UT_STAGE() dma_wait_all()

  // Fragment reads (one b128 per 32-row fragment per 8 k) into register set X or Y, and the 4*MI*NI MFMAs
  // that consume a set.
#define UT_READ(SET, buf, q)                                                                         \
  {                                                                                                  \
    const int koff_ = 4 * ((2 * (q) + fh) ^ ((fr >> 1) & 7));                                        \
    const float* as = smem + (buf) * STAGE + (wm * (MI * 32) + fr) * LDS_ROW + koff_;                \
    const float* bs = smem + (buf) * STAGE + BM * LDS_ROW + (wn * (NI * 32) + fr) * LDS_ROW + koff_; \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) af##SET[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_ROW); \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) bf##SET[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_ROW); \
  }
#define UT_MFMA_STEP(SET, C)                                                                         \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf##SET[j].C, af##SET[i].C, acc[i][j], 0, 0, 0);
  /* weights are the MFMA "A" operand, pixels the "B" operand: D = W x im2col^T, so a lane owns ONE pixel
     (column) and 4 consecutive output channels per register quad - 16-byte NHWC accesses.  The MI*NI
     accumulators are visited round-robin: consecutive MFMAs of a wave are independent (a back-to-back
     dependent 32x32x2 chain issues every ~68 cycles instead of 64 when the wave has the pipe to itself) */
#define UT_MFMA(SET)                                                                                 \
  {                                                                                                  \
    UT_MFMA_STEP(SET, x) UT_MFMA_STEP(SET, y) UT_MFMA_STEP(SET, z) UT_MFMA_STEP(SET, w)              \
  }
#define UT_PIN() __builtin_amdgcn_sched_barrier(0)
  // One 16-byte quad (G4) of accumulator (I, J) of the tile (e_tm, e_tn): (ReLU) + store through a buffer
  // descriptor (pixels beyond M and channel quads beyond cout get an out-of-range offset and are dropped).
#define UT_EPI_PART(I, J, G4)                                                                        \
  {                                                                                                  \
    const int m = e_tm * BM + wm * (MI * 32) + (I) * 32 + fr;                                        \
    const bool m_ok = m < M;                                                                         \
    const int n = e_tn * BN + wn * (NI * 32) + (J) * 32 + 8 * (G4) + 4 * fh;                         \
    float v[4];                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) v[k] = fmaxf(acc[I][J][4 * (G4) + k], e_floor);    \
    if constexpr (!NCHW) {                                                                           \
      const unsigned off = (m_ok && n < p.cout_store) ? (unsigned)(m * p.cout_store + n) * 4u + e_slab : OOB; \
      u32x4 pk;                                                                                      \
      pk.x = __float_as_uint(v[0]); pk.y = __float_as_uint(v[1]);                                    \
      pk.z = __float_as_uint(v[2]); pk.w = __float_as_uint(v[3]);                                    \
      __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, off, 0, 0);                                 \
    } else { /* NCHW (projection only): channel stride hw, one dword per channel */                  \
      const int img = fast_div(m_ok ? m : 0, hw, inv_hw);                                            \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                \
        const unsigned off = (m_ok && n + k < p.cout_store)                                          \
                                 ? (unsigned)((img * p.cout_store + n + k) * hw + (m - img * hw)) * 4u : OOB; \
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[k]), o_rsrc, off, 0, 0);             \
      }                                                                                              \
    }                                                                                                \
  }
  // The last MFMA group of a tile, one accumulator after the other instead of round-robin: the quads of the
  // accumulator that has just been finished are stored in the MFMA gaps of the next one, so that only the last
  // accumulator's stores (a quarter or a half of the epilogue) stay exposed behind the matrix work.
#define UT_Q_MFMA(SET, Q, C) acc[(Q) / NI][(Q) % NI] = __builtin_amdgcn_mfma_f32_32x32x2f32(        \
      bf##SET[(Q) % NI].C, af##SET[(Q) / NI].C, acc[(Q) / NI][(Q) % NI], 0, 0, 0)
#define UT_Q_STORE(Q, G4) if constexpr ((Q) >= 0) { UT_EPI_PART(((Q) < 0 ? 0 : (Q)) / NI, ((Q) < 0 ? 0 : (Q)) % NI, G4); }
#define UT_TAIL_Q(SET, Q)                                                                            \
  if constexpr ((Q) < MI * NI) {                                                                     \
    UT_Q_MFMA(SET, Q, x); UT_PIN(); UT_Q_STORE((Q) - 1, 0); UT_PIN();                                \
    UT_Q_MFMA(SET, Q, y); UT_PIN(); UT_Q_STORE((Q) - 1, 1); UT_PIN();                                \
    UT_Q_MFMA(SET, Q, z); UT_PIN(); UT_Q_STORE((Q) - 1, 2); UT_PIN();                                \
    UT_Q_MFMA(SET, Q, w); UT_PIN(); UT_Q_STORE((Q) - 1, 3); UT_PIN();                                \
  }
#define UT_TAIL_EPI(SET)                                                                             \
  {                                                                                                  \
    UT_TAIL_Q(SET, 0) UT_TAIL_Q(SET, 1) UT_TAIL_Q(SET, 2) UT_TAIL_Q(SET, 3)                          \
    UT_Q_STORE(MI * NI - 1, 0); UT_Q_STORE(MI * NI - 1, 1); UT_Q_STORE(MI * NI - 1, 2); UT_Q_STORE(MI * NI - 1, 3); \
  }

  // Fine interleave: one LDS-DMA piece in front of every MFMA step (a step = one k of all MI*NI accumulators),
  // instead of three bursts per chunk: a piece issued among bare MFMAs costs the issuing wave ~60 cycles, one
  // issued next to other pieces and the fragment reads 100-185 (MI355X_MICROARCH.md, LDS-DMA piece issue cost).
  // The piece's offset arithmetic and its transfer sit in different MFMA gaps (+1.1..1.5 %): in-order issue stalls the
  // wave's next MFMA only by what exceeds one 64-cycle gap.
#define UT_STEP_FINE(SET, C, IDX, DSTBUF)                                                            \
  {                                                                                                  \
    UT_PIECE_ADDR(IDX, DSTBUF); UT_PIN();                                                            \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf##SET[0].C, af##SET[0].C, acc[0][0], 0, 0, 0); \
    UT_PIN(); UT_PIECE_ISSUE(IDX); UT_PIN();                                                         \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                 \
        if (i + j > 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf##SET[j].C, af##SET[i].C, acc[i][j], 0, 0, 0); \
    UT_PIN();                                                                                        \
  }
#define UT_GROUP_FINE(SET, G, DSTBUF)                                                                \
  {                                                                                                  \
    UT_STEP_FINE(SET, x, 4 * (G) + 0, DSTBUF) UT_STEP_FINE(SET, y, 4 * (G) + 1, DSTBUF)              \
    UT_STEP_FINE(SET, z, 4 * (G) + 2, DSTBUF) UT_STEP_FINE(SET, w, 4 * (G) + 3, DSTBUF)              \
  }
#define UT_GROUP_FINE_R(SET, G, DSTBUF, READ) { READ; UT_PIN(); UT_GROUP_FINE(SET, G, DSTBUF); }
#define UT_CHUNK_FINE(buf)                                                                           \
  {                                                                                                  \
    UT_GROUP_FINE_R(X, 0, (buf) ^ 1, UT_READ(Y, buf, 1));                                            \
    UT_GROUP_FINE_R(Y, 1, (buf) ^ 1, UT_READ(X, buf, 2));                                            \
    UT_GROUP_FINE_R(X, 2, (buf) ^ 1, UT_READ(Y, buf, 3));                                            \
    UT_STAGE();                                                                       \
    UT_BARRIER();                                                                                    \
    UT_READ(X, (buf) ^ 1, 0); UT_PIN(); UT_MFMA(Y); UT_PIN();                                        \
  }
  // last chunk of a tile: the first chunk of the NEXT tile is fetched the same way, and its bias/residual requests
  // ride on the piece-free steps of the third group
#define UT_STEP_INIT(SET, C, Q, TILE) { UT_INIT_LOAD_PART(TILE, Q); UT_PIN(); UT_MFMA_STEP(SET, C) UT_PIN(); }
#define UT_CHUNK_FINE_LAST(buf, TILE)                                                                \
  {                                                                                                  \
    UT_READ(Y, buf, 1); UT_PIN(); UT_GROUP_FINE(X, 0, (buf) ^ 1);                                    \
    UT_READ(X, buf, 2); UT_PIN(); UT_GROUP_FINE(Y, 1, (buf) ^ 1);                                    \
    UT_READ(Y, buf, 3); UT_PIN();                                                                    \
    UT_STEP_INIT(X, x, 0, TILE) UT_STEP_INIT(X, y, 1, TILE) UT_STEP_INIT(X, z, 2, TILE) UT_STEP_INIT(X, w, 3, TILE) \
    UT_STAGE();                                                                       \
    UT_BARRIER();                                                                                    \
    UT_READ(X, (buf) ^ 1, 0); UT_PIN(); UT_TAIL_EPI(Y); UT_PIN();                                   \
  }
#define UT_BARRIER() __syncthreads()

  // Stagger the workgroups that share a CU.  Co-resident workgroups run the same program on the same
  // pipes; sharing the matrix pipe preserves their phase difference, and they are dispatched together, so
  // without this they stay in lockstep for the whole (persistent) kernel: every non-MFMA stretch of a chunk
  // (address math, load issue, barrier) then idles the pipe.  A one-time delay of 1/k of a chunk's MFMA time
  // per co-resident rank keeps one workgroup in its MFMA stretch while the other is between chunks.
  {
    const int rank = __builtin_amdgcn_readfirstlane((int)blockIdx.x / p.num_cu);   // 0 .. resident-1
    const int steps = rank * stagger;         // units of 512 cycles
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(8);
  }
  // Tile-queue slot: one int behind the staging area, accessed with explicit DS instructions - a `volatile int*`
  // into LDS compiles to FLAT accesses, after which every fragment wait in the chunk loop becomes lgkmcnt(0).
  const unsigned slot_addr = smem_addr + (unsigned)(2 * STAGE * 4);
#define UT_SLOT_WRITE(V) asm volatile("ds_write_b32 %0, %1" ::"v"(slot_addr), "v"(V) : "memory")
  int tile = slot;
  UT_SETUP(tile);
  UT_FETCH(0);
  UT_INIT_LOAD(tile);
  UT_STAGE();
  __syncthreads();

  int buf = 0;
  f32x16 acc[MI][NI];
  float4 afX[MI], bfX[NI], afY[MI], bfY[NI];
  UT_READ(X, 0, 0);
  for (;;) {
    UT_INIT_COMBINE();
    // the returning atomic is ISSUED here; its result goes to the LDS slot one chunk later (waiting for it here
    // would hold wave 0, and with it every chunk barrier of the workgroup, for a memory round trip per tile)
    const int ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);

    // steady state: fetch chunk c+1 (loads stay in flight under the MFMAs), compute chunk c
    for (int c = 0; c + 1 < n_chunks; ++c) {
      UT_CHUNK_FINE(buf);
      buf ^= 1;
      if (c == 0 && tid == 0) UT_SLOT_WRITE(grid + ticket);   // ordered before its read by the later chunk barriers
    }
    // The next tile comes from a device-wide queue (first round: static XCD-contiguous slots; afterwards one
    // atomic per workgroup per tile, taken a whole tile ahead by wave 0 and handed over through LDS - the chunk
    // barriers in between order it).  Dynamic hand-out keeps every CU busy when the tile count is not a multiple
    // of the resident workgroups; a static stride left up to half of them idle in the last round.
    const int e_sp = SPLITK ? tile / real_tiles : 0;
    const int e_t = tile - e_sp * real_tiles;
    const int e_tm = e_t / tiles_n, e_tn = e_t - e_tm * tiles_n;
    const unsigned e_slab = (unsigned)e_sp * (unsigned)(M * p.cout_store) * 4u;     // byte offset of the split's slab
    const float e_floor = p.relu ? 0.f : -__builtin_huge_valf();   // 0 with ReLU, -inf without: one v_max, no branch
    if (n_chunks <= 2) {                  // too few chunk barriers to order the queue slot: do it explicitly
      if (n_chunks == 1 && tid == 0) UT_SLOT_WRITE(grid + ticket);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
    }
    int next_v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(next_v) : "v"(slot_addr) : "memory");
    const int next = __builtin_amdgcn_readfirstlane(next_v);
    // Last chunk: the first chunk, bias and residual of the NEXT tile are fetched under it, and the finished
    // accumulators are stored ((ReLU) + 16-byte stores through a buffer descriptor: pixels beyond M and channel quads
    // beyond cout get an out-of-range offset and are dropped) in the MFMA gaps of its last group.  No branch: without
    // a next tile (next >= n_tiles) every row is beyond M and the requests return zeros.
    UT_SETUP(next);
    UT_CHUNK_FINE_LAST(buf, next);
    buf ^= 1;
    if ((unsigned)next >= (unsigned)n_tiles) break;     // (unsigned: a corrupt queue word cannot keep the loop alive)
    tile = next;
  }
#undef UT_SLOT_WRITE
#undef UT_SETUP
#undef UT_DECOMP
#undef UT_FETCH
#undef UT_ADVANCE
#undef UT_INIT_LOAD
#undef UT_INIT_COMBINE
#undef UT_STAGE
#undef UT_READ
#undef UT_MFMA
#undef UT_MFMA_STEP
#undef UT_PIN
#undef UT_EPI_PART
#undef UT_Q_MFMA
#undef UT_Q_STORE
#undef UT_TAIL_Q
#undef UT_TAIL_EPI
#undef UT_STEP_FINE
#undef UT_GROUP_FINE_R
#undef UT_STEP_INIT
#undef UT_CHUNK_FINE_LAST
#undef UT_INIT_LOAD_PART
#undef UT_GROUP_FINE
#undef UT_CHUNK_FINE
#undef UT_PIECE_ADDR
#undef UT_PIECE_ISSUE
#undef UT_BARRIER
}

template <int BM, int BN, int WR, int WC, bool NCHW = false, bool C32 = false, bool SPLITK = false>
static hipError_t launch_cfg(const ConvLaunch& c, hipStream_t s) {
  const int M = c.n_img * c.Ho * c.Wo;
  const int tiles_m = (M + BM - 1) / BM;
  const int tiles_n = (c.cout_store + BN - 1) / BN;
  const int n_tiles = tiles_m * tiles_n * (SPLITK ? c.splits : 1);      // queue entries
  const size_t lds = 2 * (size_t)(BM + BN) * LDS_ROW * sizeof(float) + 16;   // + tile-queue slot
  // the attribute belongs to (kernel, device): one bit per device, set on the first launch there
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WR, WC, NCHW, C32, SPLITK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  // persistent grid: as many workgroups as stay resident (LDS bound), never more than tiles
  const int per_cu = (int)((160 * 1024) / lds);
  int grid = c.num_cu * (per_cu < 1 ? 1 : per_cu);
  if (grid > n_tiles) grid = n_tiles;
  // start stagger of co-resident workgroups: with r of them a chunk takes r x (its MFMA time) of wall time, so the
  // even spacing between ranks is one chunk's MFMA time = (MI*NI) x 16 MFMAs x 64 cycles = (MI*NI) x 2 units
  const int stagger = grid > c.num_cu ? (BM / WR / 32) * (BN / WC / 32) * 2 : 0;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WR, WC, NCHW, C32, SPLITK>), dim3(grid), dim3(256), lds, s, c, tiles_n, n_tiles,
                     stagger);
  return hipGetLastError();
}

// out = act(bias + residual + slab_0 + slab_1 + ... ) in that order, 4 channels per thread (cout_store % 4 == 0)
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs, int n_splits, int mn4,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            float* __restrict__ out, int cout_store, int relu) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= mn4) return;
  const int n = (4 * i) % cout_store;
  float4 v = *reinterpret_cast<const float4*>(bias + n);
  if (res) {
    const float4 r = reinterpret_cast<const float4*>(res)[i];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  for (int sidx = 0; sidx < n_splits; ++sidx) {
    const float4 q = reinterpret_cast<const float4*>(slabs)[(size_t)sidx * mn4 + i];
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  reinterpret_cast<float4*>(out)[i] = v;
}

hipError_t launch_splitk_finish(const float* slabs, int n_splits, int m, int cout_store, const float* bias,
                                const float* res, float* out, int relu, hipStream_t s) {
  const int mn4 = m * cout_store / 4;
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((mn4 + 255) / 256), dim3(256), 0, s, slabs, n_splits, mn4, bias, res, out,
                     cout_store, relu);
  return hipGetLastError();
}

hipError_t launch_conv_igemm(const ConvLaunch& c, hipStream_t s) {
  if (c.res && c.out_nchw) return hipErrorInvalidValue;
  if (!c.tile_counter) return hipErrorInvalidValue;
  if (c.cin % 4 != 0 || c.cslice < BK || c.k_pad % BK != 0 || c.cout_pad % 128 != 0 || c.ksize * c.ksize > 9 ||
      c.num_cu <= 0)
    return hipErrorInvalidValue;
  // 32-bit byte offsets into the activation / residual tensors
  if ((size_t)c.n_img * c.H * c.W * c.cin * sizeof(float) >= 0x7FFFFF00ull) return hipErrorInvalidValue;
  if ((size_t)c.n_img * c.Ho * c.Wo * c.cout_store * sizeof(float) >= 0x7FFFFF00ull) return hipErrorInvalidValue;
  // One dispatch, by shape:
  //  layer1 (3x3 stride 1, 32 -> 32 channels)                      halo-patch kernel (conv_patch.hip)
  //  projection (the only NCHW output)                             64x128 tile
  //  backbone, cout <= 64  (channel slice == chunk width)          128x64 tile, three workgroups per CU
  //  few tiles (the head: 73,728 pixels = 576 tiles of 128 rows on 512 resident slots, i.e. two rounds the
  //  second of which is 12 % full; up to 3 x 256 full-height tiles)  64x128 tile, three workgroups per CU
  //  backbone, cout > 64 / everything else                         128x128 tile
  //  launches with fewer 64x64 tiles than CUs (a few crops)        64x64 tile
  if (c.splits > 1) {       // latency mode (ut_api.hip::run_conv): partial sums of K ranges into slabs of c.out
    if (c.out_nchw || c.res || c.relu || (c.k_pad / BK) % c.splits != 0 ||
        (size_t)c.splits * c.n_img * c.Ho * c.Wo * c.cout_store * sizeof(float) >= 0x7FFFFF00ull)
      return hipErrorInvalidValue;
    return c.cslice == BK ? launch_cfg<64, 64, 2, 2, false, true, true>(c, s) : launch_cfg<64, 64, 2, 2, false, false, true>(c, s);
  }
  const long M = (long)c.n_img * c.Ho * c.Wo;
  const bool few_tiles = ((M + 63) / 64) * ((c.cout_store + 63) / 64) <= (long)c.num_cu;
  // (latency mode also takes layer1 of a few crops off the halo-patch kernel: 144 small tiles beat 24 large ones)
  if (conv_patch_applicable(c) && !(c.splits == 1 && few_tiles)) return launch_conv_patch(c, s);
  if (c.out_nchw) return launch_cfg<64, 128, 1, 4, true>(c, s);
  const bool c32 = c.cslice == BK;          // scalar tap bookkeeping
  // a handful of crops (the per-frame tracker: 4 crops = 576 pixels at 12x12): fewer quarter-size tiles than CUs ->
  // 64x64 tiles, four times the workgroups and a quarter of the serial K walk per workgroup (same K order per output)
  if (few_tiles)
    return c32 ? launch_cfg<64, 64, 2, 2, false, true>(c, s) : launch_cfg<64, 64, 2, 2>(c, s);
  if (c.cout_store <= 64 && c32) return launch_cfg<128, 64, 2, 2, false, true>(c, s);
  const long tiles128 = ((M + 127) / 128) * ((c.cout_store + 127) / 128);
  if (tiles128 <= 3l * c.num_cu) return launch_cfg<64, 128, 1, 4>(c, s);
  return c32 ? launch_cfg<128, 128, 2, 2, false, true>(c, s) : launch_cfg<128, 128, 2, 2>(c, s);
}

}  // namespace ut
