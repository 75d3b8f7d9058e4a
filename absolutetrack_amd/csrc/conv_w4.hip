// 3x3 / stride-1 convolutions from 64 input channels up to a multiple of 128 output channels (layer3's nine 128 -> 128 and
// layer4's three 256 -> 256 convolutions, lib/models/backbone_resnet.py:56-72 at 12x12x128 and 6x6x256: 44 % of a step) in the
// split-fp16 arithmetic of conv_split.hip, as FOUR waves - one per SIMD, each with the SIMD's whole 512-entry register file - with
// NO synchronisation inside a 32-channel slice and NO address arithmetic in the MFMA stream.
//
// conv_split_kernel<256, 128, 4, 2, true> (eight waves of 64 x 64) pays ~1,600 cycles per 32-deep chunk on top of its MFMAs
// whatever their number: every chunk's weights go global -> LDS and are published to all eight waves through a counter, a wave's
// 12 MFMAs per k-step have 8 fragment reads to wait for, and every fragment read computes a swizzled, masked address.  Here
//   * a tile is 288 pixels x 128 output channels: 288 = 2 whole 12x12 maps = 8 whole 6x6 maps, so EVERY tile has the same layout
//     (nothing is recomputed per tile) and the launches are whole rounds (4096 crops: 2048 tiles = 8 per CU at 12x12, 1024 = 4 per
//     CU at 6x6; 256-pixel tiles made 4.5 rounds of layer4);
//   * a wave owns ALL 288 pixels x 32 output channels: 27 MFMAs per k-step on 18 pixel-fragment reads and TWO weight fragments;
//   * the weights never touch LDS: a wave loads the fragments of its 32 output channels straight from the fragment-ordered planes
//     into registers (one coalesced 1-KB load per fragment, no wave loads what another loads, W4_DIST k-steps ahead) - no weight
//     ring, no counter, nothing to publish;
//   * the input patch of a slice lives in LDS in PADDED image coordinates: pixel (map, y, x) at row map (H + 1)(W + 1) + y (W + 1)
//     + x (+ W + 2), the rows in between (one per image row, W + 1 per map) hold zeros - written once per workgroup, never again.
//     A tap is then a CONSTANT row shift - out-of-image taps land on the zero rows, no mask, no select - and the 16-byte groups of
//     a row are CAP rows apart (consecutive rows = consecutive 16-byte words: conflict-free without a swizzle), so a fragment read
//     is one ds_read_b128 of a per-lane base register with everything else in the instruction's immediate offset;
//   * the patch of the next slice comes global -> registers -> LDS: every thread loads nine float4, splits them into the two
//     fp16 pieces and stores the pieces at their padded position - one load, one conversion and two 8-byte LDS stores per k-step,
//     between the MFMAs; no LDS-DMA, no conversion pass;
//   * ONE barrier per slice (every 486 MFMAs of a wave): behind it the patch just written is read, the one just read rewritten;
//   * the epilogue moves residual and output as 16-byte accesses of whole 128-byte rows (each 32 x 32 block goes through 4 KB of
//     LDS from accumulator order into row order): 1 KB per instruction instead of 256 bytes - the bytes a wave can have in flight
//     behind its one 6-bit counter are what bounded it; all residual requests of a tile go out before the first value is needed.
// Every vector-memory operation is a compiler-visible load or store, so the waits are the compiler's; the order of
// a k-step's instructions is pinned slot by slot (one MFMA per slot).
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

constexpr int W4_WPIX = 288;                   // pixels of a wave: nine 32 x 32 blocks against its 32 output channels
constexpr int W4_MI = 9;
constexpr int W4_NLOAD = 9;                    // float4 patch loads per thread and slice (288 x 8 = 576 x 4 = 9 x 256)
constexpr int W4_NSTG = 5;                     // patch loads in flight per thread (loaded in k-step q, split in k-step q + 4)
constexpr int W4_DIST = 2;                     // weight fragments are loaded this many k-steps ahead
constexpr int W4_NSET = W4_DIST + 1;           // (9 and 18 k-steps % W4_NSET == 0: the register sets line up across slices)
constexpr unsigned W4_HOOB = 0x80000000u;      // out-of-range offset that stays out of range with a scalar offset added
static_assert(9 % W4_NSET == 0, "the register sets of the weight fragments line up across slices");
// padded rows of a slice patch: the tile's maps, each (H + 1)(W + 1) rows, + W + 2 rows either side, rounded up.  (The 16-byte groups of
// a pixel are then a multiple of 256 bytes apart; an odd multiple of 128 - which spreads a wave's patch stores over all banks -
// measured the same on every shape: tools/diag/w4_ab.py, W4_ALT_SRCS.)
constexpr int w4_cap(int pixels, int wi, int hi) { return ((pixels / (wi * hi)) * (hi + 1) * (wi + 1) + 2 * (wi + 2) + 15) / 16 * 16; }

// the pieces of a * s and b * s for a power of two s (conv_split.hip::split_pair_scaled)
__device__ __forceinline__ void w4_split_pair(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2w h = __builtin_bit_cast(f16x2w, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

// WI x HI: the map; SCH: input channels per slice (32: two k-steps per tap; 16: one - half the patch bytes, for maps whose padded
// tile would not fit twice otherwise); WPX: the tile's pixel halves (1: 288 pixels x 128 output channels, the four waves are its four
// 32-channel blocks; 2: 576 pixels x 64 output channels, waves (pixel half, channel block))
//
// S2: the 3x3 / stride-2 entries of layer3 and layer4 (64 -> 128 at 24x24 -> 12x12, 128 -> 256 at 12x12 -> 6x6) as PHASE PLANES.
// WI x HI is then the OUTPUT map; the input's pixels (2Y + py, 2X + px) form four planes of the output's size, and in padded
// coordinates tap (dy, dx) is a constant row shift in ONE of them: plane (1,1) carries the four corner taps (shifts (-1,-1), (-1,0),
// (0,-1), (0,0)), (0,1) the two taps dy = 0 (shifts (0,-1), (0,0)), (1,0) the two taps dx = 0 (shifts (-1,0), (0,0)) and (0,0)
// the centre.  A patch buffer holds TWO planes of 16 input channels where a stride-1 slice holds 32 channels of one (the second
// plane takes the place of the second k-step half): buffer 0 planes (1,1) + (0,0) - five k-steps -, buffer 1 planes (0,1) + (1,0) -
// four.  The nine k-steps of a 16-channel slice are one unrolled body with a barrier in front of k-steps 4 and 8; the patch stream
// is two loads and two conversions per k-step (18 per nine k-steps), every item converted four k-steps after its load, and the
// buffers never change roles.  Per output element the same products as conv_split_kernel<256, 128, 4, 2, false>, summed plane
// by plane instead of tap by tap: equal to fp32 rounding, not bit for bit.
template <int WI, int HI, int SCH, int WPX, bool S2 = false>
__global__ __launch_bounds__(256, 1) void conv_w4_kernel(ConvLaunch p, int tiles_n, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MI = W4_MI, PW = WI + 1, PH = HI + 1, HW = WI * HI;
  static_assert(!S2 || (SCH == 32 && WPX == 1), "phase planes: two 16-channel planes per buffer, four 32-channel blocks per tile");
  constexpr int BM = W4_WPIX * WPX;            // pixels of a tile: whole maps
  constexpr int NCB = 4 / WPX, BN = 32 * NCB;  // 32-channel blocks / output channels of a tile
  constexpr int NG = SCH / 4, NHG = NG / 2;    // 16-byte groups of a patch row: NHG of first pieces, then NHG of remainders
  constexpr int KS = 9 * SCH / 16;             // k-steps of a slice
  constexpr int NBUF = SCH == 32 ? 2 : 3;      // patch buffers; the patch of slice h + NBUF - 1 is fetched during slice h
  constexpr int AHEAD = NBUF - 1;
  constexpr int RP = 256 / NG;                 // pixel rows per pass of the patch stream
  constexpr int CAP = w4_cap(BM, WI, HI);      // padded rows of a slice patch
  constexpr int GSTRIDE = CAP * 16;            // bytes between two 16-byte groups of a row
  constexpr int STAGE = NG * GSTRIDE;          // one slice patch (12x12: 46 KB, 6x6: 52 KB, 24x24 in 16-channel slices: 43 KB)
  constexpr int SLOT = NBUF * STAGE;           // the next tile's index
  constexpr int EPI = SLOT + 16;               // 4 KB per wave: a 32 x 32 block on its way from accumulator to row order (epilogue)
  static_assert(BM % HW == 0, "a tile is a whole number of maps: every tile has the same padded layout");
  static_assert(BM * NG == W4_NLOAD * 256 && KS % W4_NSET == 0 && (SCH == 32 || SCH == 16) && (WPX == 1 || WPX == 2), "shape");
  static_assert((NG - 2) * GSTRIDE + (2 * PW + 2) * 16 < 65536, "fragment reads: everything but the lane's base fits the immediate offset");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WPX == 1 ? 0 : wave >> 1;     // the wave's pixel half
  const int wco = WPX == 1 ? wave : wave & 1;  // ... and 32-channel block of the tile
  const int fr = lane & 31, fh = lane >> 5;

  const int M = p.n_img * HW;
  const int n_slices = p.cin / (S2 ? 16 : SCH);      // S2: 16-channel slices of nine k-steps
  const int n_chunks = p.k_pad / 32;           // 9 per 32 input channels

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in), 0, (int)((size_t)M * (S2 ? 4 : 1) * p.cin * sizeof(float)), 0x00020000);
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

  // ---- the padded layout, the same for every tile (a tile starts on a map): pixel k of the tile - map k / HW, row y, column x -
  // sits at padded row PW + 1 + (k / HW) PH PW + y PW + x; the rows in between are zeroed once, in the prologue, and never written.
  // Patch stream: thread t loads group t % NG (four channels) of the tile's pixels RP j + t / NG, j = 0 .. 8; h_wa[j]: where their
  // first pieces go (group (t % NG) >> 1, half t & 1; the remainders NHG groups on).
  // Read side: pb[i]: the lane's pixel (of the wave's 288) 32 i + fr, + its k half as a group; taps, k-step halves and pieces are
  // immediates.
  unsigned h_wa[W4_NLOAD], pb[MI];
#pragma unroll
  for (int j = 0; j < W4_NLOAD; ++j) {
    const int k = RP * j + tid / NG, img = k / HW, rem = k - img * HW, y = rem / WI, x = rem - y * WI;
    h_wa[j] = (unsigned)((PW + 1 + img * (PH * PW) + y * PW + x) * 16 + ((tid % NG) >> 1) * GSTRIDE + (tid & 1) * 8);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int k = wm * W4_WPIX + 32 * i + fr, img = k / HW, rem = k - img * HW, y = rem / WI, x = rem - y * WI;
    pb[i] = (unsigned)((img * (PH * PW) + y * PW + x) * 16 + fh * GSTRIDE);
  }
  const unsigned h_step = (unsigned)(RP * p.cin) * 4u;
  // per-lane byte offset of pixel t / NG of a tile (pixels behind the tensor are behind the descriptor's range: they load zeros);
  // + j h_step per load, + the slice's channel offset as the scalar offset
#define W4_H_BASE(TILE) ((unsigned)((((TILE) / tiles_n) * BM + tid / NG) * p.cin + 4 * (tid % NG)) * 4u)
  // S2: the thread's group t % 8 is plane (t % 8) >> 2, channels 4 (t & 3) .. + 3 of the slice's sixteen; h_in[j]: the input pixel
  // (2Y, 2X) of its output pixel 32 j + t / 8 inside the tile's maps, po_a / po_b: its plane's pixel in buffer 0 / buffer 1
  unsigned h_in[S2 ? W4_NLOAD : 1], po_a = 0, po_b = 0, pb1[S2 ? MI : 1];
  if constexpr (S2) {
#pragma unroll
    for (int j = 0; j < W4_NLOAD; ++j) {
      const int k = RP * j + tid / NG, img = k / HW, rem = k - img * HW, y = rem / WI, x = rem - y * WI;
      h_in[j] = (unsigned)(((img * 2 * HI + 2 * y) * 2 * WI + 2 * x) * p.cin + 4 * (tid & 3)) * 4u;
    }
    const bool second = (tid & 4) != 0;
    po_a = second ? 0u : (unsigned)((2 * WI + 1) * p.cin) * 4u;
    po_b = second ? (unsigned)(2 * WI * p.cin) * 4u : (unsigned)p.cin * 4u;
#pragma unroll
    for (int i = 0; i < MI; ++i) pb1[i] = pb[i] + (unsigned)STAGE;
  }
#define W4S_BASE(TILE) ((unsigned)(((TILE) / tiles_n) * (BM * 4)) * (unsigned)p.cin * 4u)

  f32x16w acc[MI];
  u32x4w xp[MI][2];               // pixel fragments (first piece, remainder): ONE set - a fragment of the next k-step is read into its
                                  // registers as soon as this k-step's last MFMA on it has been issued
  u32x4w wf[W4_NSET][2];          // weight fragments (plane 0, plane 1) of the wave's 32 output channels
  float4 stg[S2 ? W4_NLOAD : W4_NSTG];      // patch values between their load and their split (S2: item j of either buffer in stg[j])
  const unsigned w_lane = (unsigned)lane * 16u;

  // pixel fragment PC (0 first piece, 1 remainder) of block I for k-step half S (SCH == 32) of tap TAP, base register BASE (buffer
  // included)
#define W4_READ_X(I, PC, S, TAP, BASE)                                                               \
  xp[I][PC] = *reinterpret_cast<const u32x4w*>(smem + (BASE)[I] + (unsigned)(((PC) * NHG + (SCH == 32 ? 2 * (S) : 0)) * GSTRIDE + (((TAP) / 3) * PW + (TAP) % 3) * 16));
  // weight fragment of plane PL of chunk CH, k-step half S, of the wave's block of the tile column at byte offset WROW
#define W4_LOAD_W(SET, PL, CH, S, WROW)                                                              \
  {                                                                                                  \
    const unsigned so_ = (WROW) + (unsigned)((wco * n_chunks + (CH)) * 4096 + ((S) * 2 + (PL)) * 1024); \
    wf[SET][PL] = __builtin_bit_cast(u32x4w, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, so_, 0)); \
  }
  // k-step Q of slice SL: its tap, its chunk of the weight planes and its half of the chunk
  // S2, k-step Q = 0 .. 8 of a 16-channel slice: its plane inside its buffer (buffer 1 from k-step 5 on), its shift in padded
  // coordinates ((dY + 1) * 3 + dX + 1) and the tap of the 3x3 whose weights it takes
#define W4S_PLANE(Q) ((0x190 >> (Q)) & 1)
#define W4S_TAPP(Q) ((int)((0x414344310ull >> (4 * (Q))) & 15))
#define W4S_TAPO(Q) ((int)((0x715348620ull >> (4 * (Q))) & 15))
#define W4S_BASEOF(Q) ((Q) >= 5 ? pb1 : pb)
#define W4_TAP(Q) (SCH == 32 ? (Q) >> 1 : (Q))
#define W4_CHUNK(SL, Q) (SCH == 32 ? (SL) * 9 + ((Q) >> 1) : ((SL) >> 1) * 9 + (Q))
#define W4_HALF(SL, Q) (SCH == 32 ? (Q) & 1 : (SL) & 1)
#define W4_MFMA(WSET, N)                                                                             \
  {                                                                                                  \
    constexpr int pr_ = (N) / MI, i_ = (N) % MI;                                                     \
    constexpr int wp_ = pr_ == 1 ? 1 : 0, xq_ = pr_ == 0 ? 1 : 0;      /* small terms first: x1 w0, x0 w1, x0 w0 */ \
    acc[i_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, xp[i_][xq_]),                       \
                                                     __builtin_bit_cast(f16x8w, wf[WSET][wp_]), acc[i_], 0, 0, 0);  \
  }
  // (the empty asm keeps memory operations, the scheduling barriers everything else, inside their slot)
#define W4_PIN() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define W4_ROW_OFF(I, R) ((unsigned)((I) * 32 + 8 * ((R) >> 2) + ((R) & 3)) * row_bs)
#define W4_SPLIT_STORE(V, WA)                                                                        \
  {                                                                                                  \
    unsigned a0_, b0_, a1_, b1_;                                                                     \
    w4_split_pair((V).x, (V).y, x_scale, a0_, b0_);                                                  \
    w4_split_pair((V).z, (V).w, x_scale, a1_, b1_);                                                  \
    *reinterpret_cast<u32x2w*>(smem + (WA)) = u32x2w{a0_, a1_};                                      \
    *reinterpret_cast<u32x2w*>(smem + (WA) + NHG * GSTRIDE) = u32x2w{b0_, b1_};                      \
  }

  int tile = slot;
  int next_tile = 0;
  int cur_buf = 0;                // patch buffer of the slice being computed
  unsigned out_bits = 0;
  const unsigned slot_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem + (unsigned)SLOT;

  // ---- prologue (exposed once per workgroup): the patch buffers zeroed (their zero rows stay zero for the life of the workgroup),
  // the patches of the first tile's first AHEAD slices (of the last of them, with three buffers: only what the steady state would
  // have stored by now - the rest stays in its registers for the first slice's first k-steps), the weights of the first W4_DIST
  // k-steps and the pixel fragments of the first
  for (int k = tid; k < NBUF * STAGE / 16; k += 256) *reinterpret_cast<u32x4w*>(smem + k * 16) = u32x4w{0, 0, 0, 0};
  __syncthreads();
  unsigned h_base = S2 ? 0u : W4_H_BASE(tile);
  unsigned w_row = (unsigned)((tile % tiles_n) * NCB) * (unsigned)n_chunks * 4096u;     // byte offset of the tile column's planes
  unsigned w_row_next = w_row;
  if constexpr (!S2) {
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
#pragma unroll
      for (int j = 0; j < W4_NLOAD; ++j) {
        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, h_base + (unsigned)j * h_step, (unsigned)(a * SCH * 4), 0));
        if (a == AHEAD - 1 && KS == 9 && j >= W4_NLOAD - 4) stg[j % W4_NSTG] = v;
        else W4_SPLIT_STORE(v, (unsigned)(a * STAGE) + h_wa[j])
      }
#pragma unroll
    for (int g = 0; g < W4_DIST; ++g) { W4_LOAD_W(g, 0, W4_CHUNK(0, g), W4_HALF(0, g), w_row) W4_LOAD_W(g, 1, W4_CHUNK(0, g), W4_HALF(0, g), w_row) }
  } else {
    // the state the steady state has at the start of a slice: buffer 0 complete, item 0 of buffer 1 stored, its items 1 .. 8 loaded
    const unsigned t0 = W4S_BASE(tile);
#pragma unroll
    for (int j = 0; j < W4_NLOAD; ++j) {
      const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, h_in[j] + po_a + t0, 0, 0));
      W4_SPLIT_STORE(v, h_wa[j])
    }
#pragma unroll
    for (int j = 0; j < W4_NLOAD; ++j) {
      const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, h_in[j] + po_b + t0, 0, 0));
      if (j == 0) W4_SPLIT_STORE(v, (unsigned)STAGE + h_wa[j])
      else stg[j] = v;
    }
#pragma unroll
    for (int g = 0; g < W4_DIST; ++g) { W4_LOAD_W(g, 0, W4S_TAPO(g), 0, w_row) W4_LOAD_W(g, 1, W4S_TAPO(g), 0, w_row) }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    W4_READ_X(i, 1, 0, 0, pb)
    if (i < MI - 1) { W4_READ_X(i, 0, 0, 0, pb) }      // (the last block's first piece: read in slot 0 of its own k-step)
  }

  // One slice: KS k-steps of 27 slots, one MFMA per slot (products x1 w0 of the nine pixel blocks, then x0 w1, then x0 w0).  Slot N of
  // k-step q -
  //   0             the first piece of THIS k-step's last pixel block (its registers were busy until the k-step before ended);
  //   1 .. 9        the remainder pieces of the next k-step's nine pixel blocks;   19 .. 26   the first pieces of its blocks 0 .. 7
  //                 (last k-step: of the next slice's / tile's first k-step, out of the next patch buffer);
  //   10, 11        the two weight fragments of k-step q + W4_DIST;
  //   12, 13        the split and store of the patch value loaded four k-steps ago (with 9-step slices the first four k-steps finish
  //                 the patch the slice before began);   14   a patch load of slice + AHEAD (q < 9).
#define W4_SLOT_BODY(N)                                                                              \
        {                                                                                            \
          if ((N) == 0 && q != KS - 1) { W4_READ_X(MI - 1, 0, W4_HALF(sl, q), W4_TAP(q), rb) }      /* (last k-step: in front of the barrier) */ \
          if ((N) >= 1 && (N) <= 9 && q1 < KS) { W4_READ_X(((N) + 8) % 9, 1, W4_HALF(sl, q1), W4_TAP(q1), rb) } \
          if ((N) >= 1 && (N) <= 9 && q1 == KS) { W4_READ_X(((N) + 8) % 9, 1, 0, 0, nb) }            \
          if ((N) >= 19 && (N) <= 26 && q1 < KS) { W4_READ_X(((N) + 8) % 9, 0, W4_HALF(sl, q1), W4_TAP(q1), rb) } \
          if ((N) >= 19 && (N) <= 26 && q1 == KS) { W4_READ_X(((N) + 8) % 9, 0, 0, 0, nb) }          \
          if (((N) == 10 || (N) == 11) && qd < KS) { W4_LOAD_W(qd % W4_NSET, (N) & 1, W4_CHUNK(sl, qd), W4_HALF(sl, qd), w_row) } \
          if (((N) == 10 || (N) == 11) && qd >= KS) { W4_LOAD_W(qd % W4_NSET, (N) & 1, W4_CHUNK(sl_after, qd - KS), W4_HALF(sl_after, qd - KS), row_after) } \
          if ((N) == 12 && q >= 4 && q < 4 + W4_NLOAD) { const float4 v_ = stg[(q + W4_NSTG - 4) % W4_NSTG]; w4_split_pair(v_.x, v_.y, x_scale, cv0, cv1); } \
          if ((N) == 13 && q >= 4 && q < 4 + W4_NLOAD) {                                             \
            const float4 v_ = stg[(q + W4_NSTG - 4) % W4_NSTG];                                      \
            unsigned a1_, b1_;                                                                       \
            w4_split_pair(v_.z, v_.w, x_scale, a1_, b1_);                                            \
            *reinterpret_cast<u32x2w*>(smem + wbuf + h_wa[(q + 5) % 9]) = u32x2w{cv0, a1_};          \
            *reinterpret_cast<u32x2w*>(smem + wbuf + h_wa[(q + 5) % 9] + NHG * GSTRIDE) = u32x2w{cv1, b1_}; \
          }                                                                                          \
          if (KS == 9 && (N) == 12 && q < 4) { const float4 v_ = stg[q % W4_NSTG]; w4_split_pair(v_.x, v_.y, x_scale, cv0, cv1); } \
          if (KS == 9 && (N) == 13 && q < 4) {      /* values 5 .. 8 of the patch the slice before began: the buffer before wbuf */ \
            const float4 v_ = stg[q % W4_NSTG];                                                      \
            unsigned a1_, b1_;                                                                       \
            w4_split_pair(v_.z, v_.w, x_scale, a1_, b1_);                                            \
            *reinterpret_cast<u32x2w*>(smem + pbuf + h_wa[q + 5]) = u32x2w{cv0, a1_};                \
            *reinterpret_cast<u32x2w*>(smem + pbuf + h_wa[q + 5] + NHG * GSTRIDE) = u32x2w{cv1, b1_}; \
          }                                                                                          \
          if ((N) == 14 && q < W4_NLOAD)                                                             \
            stg[q % W4_NSTG] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, h_base + (unsigned)q * h_step, f_soff, 0)); \
          if ((N) == 16 && q == KS / 2 && sl == 0 && tid == 0) *reinterpret_cast<int*>(smem + SLOT) = grid + ticket; \
          W4_PIN();                                                                                  \
          W4_MFMA(ws, N);                                                                            \
          W4_PIN();                                                                                  \
        }
#define W4_SLICE()                                                                                   \
      _Pragma("clang loop unroll(full)") for (int q = 0; q < KS; ++q) {                              \
        const int ws = q % W4_NSET;                                                                  \
        unsigned cv0 = 0, cv1 = 0;     /* the first half of the patch value being split (slot 12 -> 13) */ \
        /* (q + 1): the k-step whose pixel fragments are read now; (q + W4_DIST): the k-step whose weights are loaded now */ \
        const int q1 = q + 1, qd = q + W4_DIST;                                                      \
        if (q == KS - 1) {                                                                           \
          /* every wave has written its part of the next slice's patch and has read its last fragments of this one: ONE barrier \
             per slice, in front of the first reads of the next patch */                             \
          W4_READ_X(MI - 1, 0, W4_HALF(sl, KS - 1), 8, rb)                                           \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
          __builtin_amdgcn_s_barrier();                                                              \
          asm volatile("" ::: "memory");                                                             \
          if (sl == 0) {                                                                             \
            int nv;                                                                                  \
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nv) : "v"(slot_addr) : "memory"); \
            next_tile = __builtin_amdgcn_readfirstlane(nv);                                          \
            w_row_next = (unsigned)((next_tile % tiles_n) * NCB) * (unsigned)n_chunks * 4096u;       \
          }                                                                                          \
        }                                                                                            \
        W4_SLOT_BODY(0) W4_SLOT_BODY(1) W4_SLOT_BODY(2) W4_SLOT_BODY(3) W4_SLOT_BODY(4) W4_SLOT_BODY(5) W4_SLOT_BODY(6) W4_SLOT_BODY(7) W4_SLOT_BODY(8) \
        W4_SLOT_BODY(9) W4_SLOT_BODY(10) W4_SLOT_BODY(11) W4_SLOT_BODY(12) W4_SLOT_BODY(13) W4_SLOT_BODY(14) W4_SLOT_BODY(15) W4_SLOT_BODY(16) W4_SLOT_BODY(17) \
        W4_SLOT_BODY(18) W4_SLOT_BODY(19) W4_SLOT_BODY(20) W4_SLOT_BODY(21) W4_SLOT_BODY(22) W4_SLOT_BODY(23) W4_SLOT_BODY(24) W4_SLOT_BODY(25) W4_SLOT_BODY(26) \
      }

  // S2, k-step q of a slice: slots 0, 1 .. 9, 19 .. 26 and 10, 11 as above (the next k-step's fragments come out of buffer 1 from q = 4
  // on, out of the next slice's buffer 0 at q = 8).  The barrier of k-steps 4 and 8 sits in front of slot 9 - the k-step's own last
  // fragment was read nine MFMAs earlier, so the barrier waits for the other waves and for nothing else - with the remainder pieces
  // of the next k-step read in slots 10 .. 18.  The patch stream, item j of buffer 0 ("a") or
  // buffer 1 ("b") always through stg[j]:
  //   converted (slots 10 + 4 n .. 13 + 4 n)   q = 0 .. 3: b 2q + 1, b 2q + 2;   q = 4: a 0, a 1, a 2;   q = 5 .. 7: a 2q - 7, a 2q - 6;   q = 8: b 0
  //   loaded    (slots 22, 23, 24)             q = 0: a 0, a 1, a 2;   q = 1 .. 3: a 2q + 1, a 2q + 2;   q = 4: b 0;   q = 5 .. 8: b 2q - 9, b 2q - 8
  // - every item four k-steps after its load, an item's register free before the next item j is loaded into it; buffer 1 is written
  // from the barrier of k-step 8 (its last reader) to k-step 3, buffer 0 from the barrier of k-step 4 to k-step 7.  The "a" items
  // loaded here and the "b" items from k-step 4 on belong to the NEXT slice (or the next tile's first).  (Measured and not kept: every
  // item nine k-steps after its load, 18 registers; the weight fragments eight k-steps ahead, nine sets - both the same time.
  // Without the patch loads the launch is 11 - 13 % shorter: the 64 -> 128 entry moves its 0.9 GB at 3.2 TB/s.)
#define W4S_CV_N(Q) ((Q) == 4 ? 3 : (Q) == 8 ? 1 : 2)
#define W4S_CV_J(Q, NN) ((Q) < 4 ? 2 * (Q) + 1 + (NN) : (Q) == 4 ? (NN) : (Q) < 8 ? 2 * (Q) - 7 + (NN) : 0)
#define W4S_CV_DST(Q) (((Q) < 4 || (Q) == 8) ? (unsigned)STAGE : 0u)
#define W4S_LD_N(Q) ((Q) == 0 ? 3 : (Q) == 4 ? 1 : 2)
#define W4S_LD_J(Q, NN) ((Q) == 0 ? (NN) : (Q) < 4 ? 2 * (Q) + 1 + (NN) : (Q) == 4 ? 0 : 2 * (Q) - 9 + (NN))
#define W4S_SLOT_BODY(N)                                                                             \
        {                                                                                            \
          if ((N) == 0) { W4_READ_X(MI - 1, 0, W4S_PLANE(q), W4S_TAPP(q), W4S_BASEOF(q)) }          \
          if ((N) == 9 && (q == 4 || q == 8)) {                                                      \
            /* every wave has stored its part of the buffer read next and has read its last fragments of the other one (slot 0: nine \
               MFMAs ago - nothing is waited for but the other waves) */                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
            __builtin_amdgcn_s_barrier();                                                            \
            asm volatile("" ::: "memory");                                                           \
            if (q == 8 && sl == 0) {                                                                 \
              int nv;                                                                                \
              asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nv) : "v"(slot_addr) : "memory"); \
              next_tile = __builtin_amdgcn_readfirstlane(nv);                                        \
              w_row_next = (unsigned)((next_tile % tiles_n) * NCB) * (unsigned)n_chunks * 4096u;     \
            }                                                                                        \
          }                                                                                          \
          if ((N) >= 1 && (N) <= 9 && q != 4 && q != 8) { W4_READ_X(((N) + 8) % 9, 1, W4S_PLANE(q1), W4S_TAPP(q1), W4S_BASEOF(q1)) } \
          if ((N) >= 10 && (N) <= 18 && (q == 4 || q == 8)) { W4_READ_X(((N) + 8) % 9, 1, W4S_PLANE(q1), W4S_TAPP(q1), W4S_BASEOF(q1)) } \
          if ((N) >= 19 && (N) <= 26) { W4_READ_X(((N) + 8) % 9, 0, W4S_PLANE(q1), W4S_TAPP(q1), W4S_BASEOF(q1)) } \
          if (((N) == 10 || (N) == 11) && qd < 9) { W4_LOAD_W(qd % W4_NSET, (N) & 1, (sl >> 1) * 9 + W4S_TAPO(qd), sl & 1, w_row) } \
          if (((N) == 10 || (N) == 11) && qd >= 9) { W4_LOAD_W(qd % W4_NSET, (N) & 1, (sl_after >> 1) * 9 + W4S_TAPO(qd - 9), sl_after & 1, row_after) } \
          /* a conversion in four slots of four vector instructions: (x, y) first pieces, their remainders, (z, w) likewise + the stores */ \
          if ((N) >= 10 && (N) <= 21 && ((N) - 10) / 4 < W4S_CV_N(q)) {                              \
            const float4 v_ = stg[W4S_CV_J(q, ((N) - 10) / 4) % 9];                                  \
            constexpr int part_ = ((N) - 10) & 3;                                                    \
            if (part_ == 0 || part_ == 2) {                                                          \
              const f16x2w h_ = __builtin_bit_cast(f16x2w, __builtin_amdgcn_cvt_pkrtz((part_ ? v_.z : v_.x) * x_scale, (part_ ? v_.w : v_.y) * x_scale)); \
              (part_ ? cvh1 : cvh0) = __builtin_bit_cast(unsigned, h_);                              \
              cvf = (float)h_[0];                                                                    \
            } else {                                                                                 \
              const f16x2w h_ = __builtin_bit_cast(f16x2w, part_ == 1 ? cvh0 : cvh1);                \
              const float ra_ = __builtin_fmaf(part_ == 1 ? v_.x : v_.z, x_scale, -cvf);             \
              const float rb_ = __builtin_fmaf(part_ == 1 ? v_.y : v_.w, x_scale, -(float)h_[1]);    \
              const unsigned p_ = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra_, rb_)); \
              if (part_ == 1) cvp0 = p_;                                                             \
              else {                                                                                 \
                *reinterpret_cast<u32x2w*>(smem + W4S_CV_DST(q) + h_wa[W4S_CV_J(q, ((N) - 10) / 4) % 9]) = u32x2w{cvh0, cvh1}; \
                *reinterpret_cast<u32x2w*>(smem + W4S_CV_DST(q) + h_wa[W4S_CV_J(q, ((N) - 10) / 4) % 9] + NHG * GSTRIDE) = u32x2w{cvp0, p_}; \
              }                                                                                      \
            }                                                                                        \
          }                                                                                          \
          if ((N) >= 22 && (N) <= 24 && (N) - 22 < W4S_LD_N(q))                                      \
            stg[W4S_LD_J(q, (N) - 22) % 9] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(    \
                a_rsrc, h_in[W4S_LD_J(q, (N) - 22) % 9] + (q < 4 ? hb_a : hb_b), f_soff, 0));        \
          if ((N) == 16 && q == 1 && sl == 0 && tid == 0) *reinterpret_cast<int*>(smem + SLOT) = grid + ticket; \
          W4_PIN();                                                                                  \
          W4_MFMA(ws, N);                                                                            \
          W4_PIN();                                                                                  \
        }
#define W4S_SLICE()                                                                                  \
      _Pragma("clang loop unroll(full)") for (int q = 0; q < 9; ++q) {                               \
        const int ws = q % W4_NSET;                                                                  \
        unsigned cvh0 = 0, cvh1 = 0, cvp0 = 0;     /* the conversion in progress */                  \
        float cvf = 0.f;                                                                             \
        const int q1 = (q + 1) % 9, qd = q + W4_DIST;                                                \
        W4S_SLOT_BODY(0) W4S_SLOT_BODY(1) W4S_SLOT_BODY(2) W4S_SLOT_BODY(3) W4S_SLOT_BODY(4) W4S_SLOT_BODY(5) W4S_SLOT_BODY(6) W4S_SLOT_BODY(7) W4S_SLOT_BODY(8) \
        W4S_SLOT_BODY(9) W4S_SLOT_BODY(10) W4S_SLOT_BODY(11) W4S_SLOT_BODY(12) W4S_SLOT_BODY(13) W4S_SLOT_BODY(14) W4S_SLOT_BODY(15) W4S_SLOT_BODY(16) W4S_SLOT_BODY(17) \
        W4S_SLOT_BODY(18) W4S_SLOT_BODY(19) W4S_SLOT_BODY(20) W4S_SLOT_BODY(21) W4S_SLOT_BODY(22) W4S_SLOT_BODY(23) W4S_SLOT_BODY(24) W4S_SLOT_BODY(25) W4S_SLOT_BODY(26) \
      }

  for (;;) {
    // the tile after this one: the ticket is taken here, written to LDS by thread 0 in the middle of the tile's first slice and
    // read by everyone behind that slice's barrier (a tile has at least AHEAD + 1 slices)
    int ticket = 0;
    if (tid == 0) ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // (the tile's bias, for the epilogue: requested here, long before it is needed)
    const float4 bb4 = *reinterpret_cast<const float4*>(p.bias + (tile % tiles_n) * BN + wco * 32 + 4 * (lane & 7));
    if constexpr (S2) {
      for (int sl = 0; sl < n_slices; ++sl) {
        const bool last_slice = sl == n_slices - 1;
        // the patch items loaded during this slice: the next slice's, or the next tile's first (a tile has at least two slices: the
        // next tile is known from the first slice's second barrier on)
        const unsigned f_soff = last_slice ? 0u : (unsigned)(sl + 1) * 64u;
        const unsigned f_tb = W4S_BASE(last_slice ? next_tile : tile);
        const unsigned hb_a = po_a + f_tb, hb_b = po_b + f_tb;
        const int sl_after = last_slice ? 0 : sl + 1;
        const unsigned row_after = last_slice ? w_row_next : w_row;
        W4S_SLICE()
      }
    } else
    for (int sl = 0; sl < n_slices; ++sl) {
      const bool last_slice = sl == n_slices - 1;
      // the buffers: read this slice's patch, the next slice's (its first fragments are read in this slice's last k-step), the
      // patch being begun (slice + AHEAD) and - three buffers - the one the slice before began and this slice finishes
      const int b1 = cur_buf + 1 == NBUF ? 0 : cur_buf + 1, b2 = b1 + 1 == NBUF ? 0 : b1 + 1;
      unsigned rbuf = (unsigned)__builtin_amdgcn_readfirstlane(cur_buf * STAGE), nbuf = (unsigned)__builtin_amdgcn_readfirstlane(b1 * STAGE);
      unsigned wbuf = (unsigned)__builtin_amdgcn_readfirstlane((NBUF == 2 ? b1 : b2) * STAGE), pbuf = nbuf;
      asm volatile("" : "+s"(rbuf), "+s"(nbuf), "+s"(wbuf), "+s"(pbuf));      // (opaque per slice: nothing derived from them is carried across slices)
      // the patch begun during this slice: slice + AHEAD of this tile, or of the next
      const int f_sl = sl + AHEAD < n_slices ? sl + AHEAD : sl + AHEAD - n_slices;
      if (sl + AHEAD == n_slices) h_base = W4_H_BASE(next_tile);
      const unsigned f_soff = (unsigned)f_sl * (SCH * 4);
      // the weight fragments of the k-steps behind this slice: the next slice's, or the next tile's first
      const int sl_after = last_slice ? 0 : sl + 1;
      const unsigned row_after = last_slice ? w_row_next : w_row;
      unsigned rb[MI], nb[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) { rb[i] = pb[i] + rbuf; nb[i] = pb[i] + nbuf; }
      W4_SLICE()
      cur_buf = b1;
    }
    // ---- epilogue: 1 / (weight scale x activation scale) x accumulator + bias + residual, ReLU, store.  The pixels are the
    // MFMAs' first operand, so a lane's sixteen registers of a block are ONE output channel of sixteen pixels and one dword access
    // per register covers two whole 128-byte half rows (conv_split.hip); a block's sixteen rows are one per-lane base plus
    // WAVE-UNIFORM row offsets in the instruction's scalar offset (one offset register per wave, not one per access).  All 144
    // residual requests of a wave go out before the first value is needed: one memory latency per tile, not one per block.  The
    // scalar offset is not part of the descriptor's range check: the tile that reaches beyond the tensor takes per-access offsets.
    {
      const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
      const bool ragged = tm * BM + BM > M;
      const int ch = tn * BN + wco * 32 + fr;
      const float bb = p.bias[ch];
      const unsigned base = ch < p.cout_store ? (unsigned)((tm * BM + wm * W4_WPIX + 4 * fh) * p.cout_store + ch) * 4u : W4_HOOB;
      const unsigned keep_n = base != W4_HOOB ? 0x7FFFFFFFu : 0u;
      if (!ragged) {
        // A lane's sixteen accumulator registers of a block are one channel of sixteen pixels: stored as they stand, an instruction
        // moves 256 bytes (two half rows), and with 63 of them in flight a wave keeps 16 KB on its way - a CU then moves ~30 GB/s
        // whatever the rest of the chip does, and a tile's 294 KB of residual and output took a sixth to a third of its time.
        // So every block goes through 4 KB of LDS into ROW order - lane l holds pixel 8 t + (l >> 3), channels 4 (l & 7) .. + 3 -
        // and residual and output move as 16-byte accesses: 1 KB (eight whole 128-byte rows) per instruction, four times the
        // bytes in flight.  Same arithmetic per element.
        unsigned row_bs = (unsigned)__builtin_amdgcn_readfirstlane((int)row_b);
        asm volatile("" : "+s"(row_bs));
        const int ch4 = tn * BN + wco * 32 + 4 * (lane & 7);
        const unsigned base4 = ch4 < p.cout_store ? (unsigned)((tm * BM + wm * W4_WPIX + (lane >> 3)) * p.cout_store + ch4) * 4u : W4_HOOB;
        const unsigned keep4 = base4 != W4_HOOB ? 0x7FFFFFFFu : 0u;
        // (the residual of five blocks is requested up front, that of block i + 5 at the start of block i - in front of block i's
        // stores, so that waiting for it never waits for a store younger than five blocks: 96 registers instead of 144.  Without a
        // residual the descriptor has no records: the requests return zeros without touching memory.)
        // (S2: the stride-2 entries have no residual - and eight patch items in flight across the epilogue where stride 1 has none)
        float4 rr[S2 ? 1 : 6][4];
        if constexpr (!S2) {
#pragma unroll
          for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              rr[i][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, base4, (unsigned)(32 * i + 8 * t) * row_bs, 0));
        }
        // half a block (16 pixels x 32 channels, 2 KB) at a time, two regions in turn: the writes of one half go out while the reads
        // of the half before are on their way back
        char* const ex = smem + EPI + wave * 4096;
#define W4_EX_WRITE(H)                                                                               \
        _Pragma("unroll") for (int r = 0; r < 8; ++r)                                                \
          *reinterpret_cast<float*>(ex + ((H) & 1) * 2048 + (8 * (r >> 2) + 4 * fh + (r & 3)) * 128 + fr * 4) = acc[(H) >> 1][8 * ((H) & 1) + r];
        W4_EX_WRITE(0)
#pragma unroll
        for (int h = 0; h < 2 * MI; ++h) {
          const int i = h >> 1;
          if (!S2 && (h & 1) == 0 && i + 5 < MI) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              rr[(i + 5) % 6][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, base4, (unsigned)(32 * (i + 5) + 8 * t) * row_bs, 0));
          }
          if (h + 1 < 2 * MI) { W4_EX_WRITE(h + 1) }
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int t = 2 * (h & 1) + t2;
            const float4 a = *reinterpret_cast<const float4*>(ex + (h & 1) * 2048 + (8 * t2 + (lane >> 3)) * 128 + (lane & 7) * 16);
            const float4 res = S2 ? float4{0.f, 0.f, 0.f, 0.f} : rr[S2 ? 0 : i % 6][t];
            u32x4w o;
            o.x = __float_as_uint(fmaxf(fmaf(a.x, tot_unscale, bb4.x + res.x), floor_v));
            o.y = __float_as_uint(fmaxf(fmaf(a.y, tot_unscale, bb4.y + res.y), floor_v));
            o.z = __float_as_uint(fmaxf(fmaf(a.z, tot_unscale, bb4.z + res.z), floor_v));
            o.w = __float_as_uint(fmaxf(fmaf(a.w, tot_unscale, bb4.w + res.w), floor_v));
            unsigned mk;      // (an asm max: as a plain max the compiler builds one reduction tree and keeps every value alive for it)
            asm volatile("v_max3_u32 %0, %2, %3, %4\n\tv_max_u32 %0, %0, %5\n\tv_and_b32 %0, %0, %6\n\tv_max_u32 %1, %1, %0"
                         : "=&v"(mk), "+v"(out_bits) : "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w), "v"(keep4));
            __builtin_amdgcn_raw_buffer_store_b128(o, o_rsrc, base4, (unsigned)(32 * i + 8 * t) * row_bs, 0);
          }
        }
#undef W4_EX_WRITE
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          float r1[16];
          unsigned off = base + (unsigned)(i * 32) * row_b;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            r1[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off, 0, 0));
            off += ((r & 3) == 3 ? 5u : 1u) * row_b;
          }
          off = base + (unsigned)(i * 32) * row_b;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned o = __float_as_uint(fmaxf(fmaf(acc[i][r], tot_unscale, bb + r1[r]), floor_v));
            const int pix = wm * W4_WPIX + i * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
            out_bits = max(out_bits, o & (tm * BM + pix < M ? keep_n : 0u));
            __builtin_amdgcn_raw_buffer_store_b32(o, o_rsrc, off, 0, 0);
            off += ((r & 3) == 3 ? 5u : 1u) * row_b;
          }
        }
      }
    }
    if ((unsigned)next_tile >= (unsigned)n_tiles) break;
    tile = next_tile;
    w_row = w_row_next;
  }
  if (p.out_max) publish_abs_max(p.out_max, out_bits);
#undef W4_H_BASE
#undef W4_READ_X
#undef W4_LOAD_W
#undef W4_TAP
#undef W4_CHUNK
#undef W4_HALF
#undef W4_MFMA
#undef W4_PIN
#undef W4_ROW_OFF
#undef W4_SPLIT_STORE
#undef W4_SLOT_BODY
#undef W4_SLICE
#undef W4S_PLANE
#undef W4S_TAPP
#undef W4S_TAPO
#undef W4S_BASEOF
#undef W4S_BASE
#undef W4S_CV_N
#undef W4S_CV_J
#undef W4S_CV_DST
#undef W4S_LD_N
#undef W4S_LD_J
#undef W4S_SLOT_BODY
#undef W4S_SLICE
}

template <int WI, int HI, int SCH, int WPX, bool S2 = false>
hipError_t launch_w4_cfg(const ConvLaunch& c, hipStream_t s) {
  constexpr int BM = W4_WPIX * WPX, BN = 128 / WPX;
  const long M = (long)c.n_img * c.Ho * c.Wo;
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = c.cout_store / BN;
  const int n_tiles = tiles_m * tiles_n;
  constexpr int lds = (SCH == 32 ? 2 : 3) * (SCH / 4) * w4_cap(BM, WI, HI) * 16 + 16 + 4 * 4096;
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_w4_kernel<WI, HI, SCH, WPX, S2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL((conv_w4_kernel<WI, HI, SCH, WPX, S2>), dim3(grid), dim3(256), lds, s, c, tiles_n, n_tiles);
  return hipGetLastError();
}

}  // namespace

// (the map sizes are template parameters - taps and padded rows are immediates of the fragment reads: the backbone's 12x12 and 6x6 maps
// with 128-channel tiles, its 24x24 maps with 64 -> 64 channels as tiles of one whole map; stride 2: the entries of layer3 and layer4,
// 12x12 and 6x6 OUTPUT maps)
bool conv_w4_applicable(const ConvLaunch& c) {
  const bool common = !(c.no_resident & 2) && c.w_split && c.split_unscale > 0.f && c.ksize == 3 && c.pad == 1 && c.cslice == 32 &&
                      c.cin % 32 == 0 && c.cout_pad >= c.cout_store && !c.out_nchw && c.splits == 0 &&
                      c.k_pad == 9 * c.cin && c.tile_counter && c.num_cu > 0 &&
                      (size_t)c.n_img * c.H * c.W * c.cin * sizeof(float) < 0x7FFFFF00ull &&
                      (size_t)c.n_img * c.Ho * c.Wo * c.cout_store * sizeof(float) < 0x7FFFFF00ull;
  if (!common) return false;
  if (c.stride == 2)      // phase planes
    return !(c.no_resident & 8) && c.cin >= 64 /* (what the tests cover: 64 and 128; two-slice tiles are untested) */ && c.cout_store % 128 == 0 && c.H == 2 * c.Ho && c.W == 2 * c.Wo && !c.res &&
           ((c.Wo == 12 && c.Ho == 12) || (c.Wo == 6 && c.Ho == 6));
  if (c.stride != 1 || c.cin < 64 || c.H != c.Ho || c.W != c.Wo) return false;
  if (c.cout_store % 128 == 0) return (c.W == 12 && c.H == 12) || (c.W == 6 && c.H == 6);
  return !(c.no_resident & 4) && c.cout_store == 64 && c.W == 24 && c.H == 24;
}

hipError_t launch_conv_w4(const ConvLaunch& c, hipStream_t s) {
  if (!conv_w4_applicable(c)) return hipErrorInvalidValue;
  if (c.stride == 2) return c.Wo == 12 ? launch_w4_cfg<12, 12, 32, 1, true>(c, s) : launch_w4_cfg<6, 6, 32, 1, true>(c, s);
  if (c.W == 24) return launch_w4_cfg<24, 24, 16, 2>(c, s);
  return c.W == 12 ? launch_w4_cfg<12, 12, 32, 1>(c, s) : launch_w4_cfg<6, 6, 32, 1>(c, s);
}

}  // namespace ut
