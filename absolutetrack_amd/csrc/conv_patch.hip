// 3x3 / stride-1 / 32->32-channel convolution (the four convolutions of ResNet layer1,
// lib/models/backbone_resnet.py:56-72 at 48x48x32) with the input HALO PATCH resident in LDS.
//
// In the generic implicit GEMM (conv_igemm.hip) every one of the 9 taps re-stages the same pixels through
// LDS: with only 32 output channels that is 6.5 MAC per staged byte and the global->LDS path (not the MFMA
// pipe, not latency) is what bounds it at ~60 % of the fp32-MFMA rate.  Here one workgroup (12 waves, one
// per CU) owns a 16x24-pixel output tile: the 18x26x32 input patch is brought into LDS ONCE by LDS-DMA
// (double buffered: the next tile's patch streams in under the current tile's MFMAs), all 9x32x32 weights stay
// resident in LDS for the life of the (persistent) workgroup, and the 9 taps read shifted windows of the patch.
// Global->LDS traffic drops from 270 KB to 60 KB per 384 output pixels; there is no barrier inside a tile's
// 144-MFMA stream, one per tile.
//
// MFMA operand roles as in conv_igemm.hip: weights = "A" (rows = output channel), pixels = "B" (columns), so a
// lane owns one pixel and accumulator register quads are 4 consecutive channels (16-byte NHWC accesses).
// LDS rows are 128 B (32 floats) with the 16-byte chunks XOR-swizzled by ((row >> 1) & 7), applied on the source
// side of the DMA; rows are "pixel of the patch" resp. "(tap, output channel)".
#include <atomic>

#include "ut_kernels.h"

namespace ut {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32p;

namespace {
constexpr int TY = 16, TX = 24, NW = 12;    // output tile is TY x TX pixels, one 32-pixel block per wave: 12 waves,
                                            // three per SIMD (+3.7 % over two), 157.7 KB of the CU's 160 KB LDS
static_assert(TY * TX == 32 * NW, "one 32-pixel block per wave");
constexpr int PW = TX + 2, PH = TY + 2;     // patch width / height
constexpr int PPIX = PW * PH;               // 324 (468) patch pixels
constexpr int PROWS = (PPIX + 7) / 8 * 8;   // 328 (472): DMA pieces cover whole 8-row groups
constexpr int PDIV = (65536 + PW - 1) / PW; // pidx / PW == (pidx * PDIV) >> 16 for pidx < PROWS (checked on the host)
constexpr int C = 32;
constexpr int W_FLOATS = 9 * C * C;         // 9216
constexpr int P_FLOATS = PROWS * C;         // 10496
constexpr int PATCH_INSTR = PROWS / 8;      // 41 wave-level DMA instructions per patch
constexpr int W_INSTR = 9 * C / 8;          // 36
constexpr unsigned OOBP = 0xFFFFFF00u;

__device__ __forceinline__ void dma16p(u32x4 rsrc, unsigned lds_addr, unsigned voffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc)
      : "memory");
}
__device__ __forceinline__ u32x4 rsrc_words(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
}  // namespace

// SPLIT: the arithmetic of conv_split.hip (two fp16 pieces per operand, three products per k on v_mfma_f32_32x32x16_f16,
// fp32 accumulation) instead of the fp32 matrix instruction: the resident weights are the two fp16 planes of
// ConvLaunch::w_split in fragment order (the same 36 KB), the patch stays fp32 in LDS and is split in registers, the
// accumulators hold scale x (sum) and start at scale x (bias + residual).  Nine taps x 6 MFMAs of 32 cycles instead of
// 9 x 16 of 64: the kernel is then bound by the tile's HBM traffic (patch in, residual in, tile out), not by the matrix pipe.
// The activations are multiplied by a power of two taken from the producer's max word before they are split
// (ut_kernels.h::split_act_scale), so the split has no precondition on their magnitude; the kernel leaves its own max word.
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_p(float a, float b, unsigned& p0, unsigned& p1) {
  const f16x2p h = __builtin_bit_cast(f16x2p, __builtin_amdgcn_cvt_pkrtz(a, b));
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

__device__ __forceinline__ void split_pair_scaled_p(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2p h = __builtin_bit_cast(f16x2p, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);   // exact products: see conv_split.hip
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

template <bool SPLIT>
__global__ __launch_bounds__(64 * NW) void conv3x3_c32_patch_kernel(ConvLaunch p, int tiles_x, int tiles_per_img, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w_lds = smem;                                  // [9*32 rows][32]
  float* patch0 = smem + W_FLOATS;                      // 2 x [328 rows][32]
  // tile-queue hand-over words behind the patches.  Accessed with explicit DS instructions: a `volatile int*`
  // into LDS is compiled as a FLAT load, which counts on vmcnt and would wait for the tile's stores.
  const unsigned slot_addr0 = (unsigned)(unsigned long)(lds_f32p*)smem + (unsigned)((W_FLOATS + 2 * P_FLOATS) * 4);
  auto slot_write = [&](int idx, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(slot_addr0 + 4u * (unsigned)idx), "v"(v) : "memory");
  };
  auto slot_read = [&](int idx) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(slot_addr0 + 4u * (unsigned)idx) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int fr = lane & 31, fh = lane >> 5;
  const int H = p.H, W = p.W;
  const int M = p.n_img * H * W;
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_f32p*)smem;

  // SPLIT: a landed patch (fp32 rows, 16-byte group g at position g ^ swizzle) is split IN PLACE, once per tile - a value's
  // two fp16 pieces take its 4 bytes; group q = 4 * piece + k / 8 at position q ^ swizzle.  Thread t < PROWS owns row t.
  // The nine taps then read ready pieces instead of splitting the same values nine times.
  float x_scale = 1.f, x_unscale = 1.f;      // SPLIT: power-of-two activation scale and its inverse
  if constexpr (SPLIT) {
    if (p.in_max) {
      bool ok;
      split_act_scale(p.in_max, p.in_obs, x_scale, x_unscale, ok);
      if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);
    }
  }
  auto convert_patch = [&](int buf) {
    if (tid < PROWS) {
      const int sw = (tid >> 1) & 7;
      char* rp = reinterpret_cast<char*>(smem) + (size_t)(W_FLOATS + buf * P_FLOATS) * 4 + tid * 128;
      float4 f[8];
#pragma unroll
      for (int g4 = 0; g4 < 8; ++g4) f[g4] = *reinterpret_cast<const float4*>(rp + ((g4 ^ sw) << 4));
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        unsigned a0, a1, a2, a3, b0, b1, b2, b3;
        split_pair_scaled_p(f[2 * kg].x, f[2 * kg].y, x_scale, a0, b0);
        split_pair_scaled_p(f[2 * kg].z, f[2 * kg].w, x_scale, a1, b1);
        split_pair_scaled_p(f[2 * kg + 1].x, f[2 * kg + 1].y, x_scale, a2, b2);
        split_pair_scaled_p(f[2 * kg + 1].z, f[2 * kg + 1].w, x_scale, a3, b3);
        u32x4 a, b;
        a.x = a0; a.y = a1; a.z = a2; a.w = a3;
        b.x = b0; b.y = b1; b.z = b2; b.w = b3;
        *reinterpret_cast<u32x4*>(rp + ((kg ^ sw) << 4)) = a;
        *reinterpret_cast<u32x4*>(rp + (((4 + kg) ^ sw) << 4)) = b;
      }
    }
  };

  const u32x4 in_words = rsrc_words(p.in, (unsigned)((size_t)M * C * sizeof(float)));
  const u32x4 w_words = SPLIT ? rsrc_words(p.w_split, (unsigned)(W_FLOATS * 4))
                              : rsrc_words(p.w, (unsigned)((size_t)p.cout_pad * p.k_pad * sizeof(float)));
  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.bias), 0, p.res ? (int)((size_t)M * C * sizeof(float)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)M * C * sizeof(float)), 0x00020000);

  // ---- weights -> LDS once (row = tap*32 + n, swizzle by n)
  for (int k = wave; k < W_INSTR; k += NW) {
    if constexpr (SPLIT) {     // rows 0..31 of w_split: [tap][k-step][plane][lane][8 halves], one 1-KB block per piece
      dma16p(w_words, smem_addr + (unsigned)(k * 1024), (unsigned)(k * 1024 + lane * 16));
      continue;
    }
    const int e = k * 64 + lane;
    const int row = e >> 3, cpos = e & 7;
    const int n = row & 31, tap = row >> 5;
    const int c4 = cpos ^ ((n >> 1) & 7);
    dma16p(w_words, smem_addr + (unsigned)(k * 64 * 16), (unsigned)((n * p.k_pad + tap * C + 4 * c4) * 4));
  }

  // ---- per-lane constants of the patch DMA: which patch pixel / channel chunk each of my pieces is
  constexpr int MAXP = (PATCH_INSTR + NW - 1) / NW;   // 6 (5) instructions per wave at most
  int pc_py[MAXP], pc_px[MAXP], pc_c4[MAXP];
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int k = wave + NW * j;
    const int e = k * 64 + lane;
    const int pidx = e >> 3, cpos = e & 7;
    const int py = (pidx * PDIV) >> 16;       // pidx / PW
    pc_py[j] = (k < PATCH_INSTR && pidx < PPIX) ? py : -1000;
    pc_px[j] = pidx - py * PW;
    pc_c4[j] = cpos ^ ((pidx >> 1) & 7);
  }

  // tile -> (first output pixel row index of the image, y0, x0); scalar float-reciprocal division (tile counts
  // are far below 2^24)
  const float inv_tpi = 1.0f / (float)tiles_per_img, inv_tx = 1.0f / (float)tiles_x;
  auto tile_origin = [&](int tile, int& row0, int& y0, int& x0) {
    int img = (int)((float)tile * inv_tpi);
    int r = tile - img * tiles_per_img;
    if (r < 0) { --img; r += tiles_per_img; }
    if (r >= tiles_per_img) { ++img; r -= tiles_per_img; }
    int ty = (int)((float)r * inv_tx);
    int c = r - ty * tiles_x;
    if (c < 0) { --ty; c += tiles_x; }
    if (c >= tiles_x) { ++ty; c -= tiles_x; }
    row0 = img * H;
    y0 = ty * TY;
    x0 = c * TX;
  };

  // one wave-level DMA instruction (j-th of this wave) of the patch of the tile with origin (row0, y0, x0)
  auto issue_piece = [&](int j, int row0, int y0, int x0, int buf) {
    const int k = wave + NW * j;
    if (k < PATCH_INSTR) {     // wave-uniform
      const int gy = y0 - 1 + pc_py[j], gx = x0 - 1 + pc_px[j];
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)((((row0 + gy) * W + gx) * C + 4 * pc_c4[j]) * 4) : OOBP;
      dma16p(in_words, smem_addr + (unsigned)((W_FLOATS + buf * P_FLOATS) * 4 + k * 64 * 16), off);
    }
  };

  // my output pixel inside the tile and its position in the patch (tap 0,0 = one up, one left)
  // my output pixel inside the tile: the wave's 32 pixels are consecutive in the tile's raster order
  const int lq = 32 * wave + fr;
  const int ly = TX == 16 ? (lq >> 4) : lq / TX, lx = lq - ly * TX;
  const int pbase = ly * PW + lx;
  // weight fragment: row = tap*32 + fr (output channel fr), chunk (2q+fh) swizzled by fr
  int w_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w_off[q] = fr * C + 4 * ((2 * q + fh) ^ ((fr >> 1) & 7));

  // bias and residual of a tile are REQUESTED a tile ahead and only combined when needed: any arithmetic on
  // them at request time would make the compiler wait for the loads in front of the MFMAs
  u32x4 res_raw[4];
  float4 bias_raw[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) bias_raw[g4] = *reinterpret_cast<const float4*>(p.bias + 8 * g4 + 4 * fh);
  auto init_load = [&](int row0, int y0, int x0) {
    const int m = (row0 + y0 + ly) * W + x0 + lx;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      res_raw[g4] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, (unsigned)(m * C + 8 * g4 + 4 * fh) * 4u, 0, 0);
  };
  const float acc_scale = SPLIT ? x_scale / p.split_unscale : 1.0f;      // a power of two: (weight scale) x (activation scale)
  const float acc_unscale = SPLIT ? p.split_unscale * x_unscale : 1.0f;
  auto init_combine = [&]() {
    f32x16 v;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      v[4 * g4 + 0] = bias_raw[g4].x + __uint_as_float(res_raw[g4].x);
      v[4 * g4 + 1] = bias_raw[g4].y + __uint_as_float(res_raw[g4].y);
      v[4 * g4 + 2] = bias_raw[g4].z + __uint_as_float(res_raw[g4].z);
      v[4 * g4 + 3] = bias_raw[g4].w + __uint_as_float(res_raw[g4].w);
    }
    if constexpr (SPLIT) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] *= acc_scale;
    }
    return v;
  };
  // tile-queue ticket: a straight-line buffer atomic (only thread 0 has an in-range offset; the others are
  // dropped by the range check) so that its result is awaited where it is used, not where it is issued
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.tile_counter, 0, 4, 0x00020000);
  const unsigned q_off = tid == 0 ? 0u : OOBP;

  const int grid = gridDim.x;
  int tile = blockIdx.x;
  int c_row0, c_y0, c_x0;            // origin of the tile being computed
  tile_origin(tile, c_row0, c_y0, c_x0);
#pragma unroll
  for (int j = 0; j < MAXP; ++j) issue_piece(j, c_row0, c_y0, c_x0, 0);
  init_load(c_row0, c_y0, c_x0);
  {
    const int t0 = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);
    if (tid == 0) slot_write(2, grid + t0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (SPLIT) {
    convert_patch(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  int next = slot_read(2);
  int cur = 0;

#define UTP_EPILOGUE(M_)                                                                              \
  {                                                                                                  \
    const int m_ = (M_);                                                                             \
    _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                               \
      const int n = 8 * g4 + 4 * fh;                                                                 \
      u32x4 pk;                                                                                      \
      float v0 = acc[4 * g4], v1 = acc[4 * g4 + 1], v2 = acc[4 * g4 + 2], v3 = acc[4 * g4 + 3];      \
      if constexpr (SPLIT) { v0 *= acc_unscale; v1 *= acc_unscale; v2 *= acc_unscale; v3 *= acc_unscale; } \
      if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); } \
      if constexpr (SPLIT) out_bits = max(max(out_bits, max(abs_bits(v0), abs_bits(v1))), max(abs_bits(v2), abs_bits(v3))); \
      pk.x = __float_as_uint(v0); pk.y = __float_as_uint(v1); pk.z = __float_as_uint(v2); pk.w = __float_as_uint(v3); \
      __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, (unsigned)(m_ * C + n) * 4u, 0, 0);         \
    }                                                                                                \
  }
  const bool late = ((wave >> 2) & 1) != 0;   // waves w, w+4(, w+8) share a SIMD; wave is uniform (readfirstlane)
  bool have_prev = false;
  int prev_m = 0;
  unsigned out_bits = 0;    // SPLIT: bits of the largest output magnitude of this lane (the next layer scales by it before splitting)
  f32x16 ready = init_combine();      // bias + residual of the tile about to be computed
  f32x16 acc;
  for (;;) {
    const bool has_next = (unsigned)next < (unsigned)n_tiles;     // (unsigned: a corrupt queue word cannot keep the loop alive)
    if (late && have_prev) { UTP_EPILOGUE(prev_m); }
    acc = ready;
    // ticket for the tile after next: issued now, consumed before the last tap
    const int ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);
    int n_row0 = 0, n_y0 = 0, n_x0 = 0;
    if (has_next) tile_origin(next, n_row0, n_y0, n_x0);

    const float* patch = patch0 + cur * P_FLOATS;
    // 9 taps x 4 k-groups, fragments double buffered in registers.  The requests for the NEXT tile (patch
    // pieces, residual) are slipped in between MFMA groups of taps 0..6 so that their address arithmetic and
    // issue run in the shadow of MFMAs already queued.
    if constexpr (!SPLIT) {
      float4 wfX, pfX, wfY, pfY;
#define UTP_READ(SET, TAP, Q)                                                                        \
    {                                                                                                  \
      const int pidx_ = pbase + ((TAP) / 3) * PW + ((TAP) % 3);                                        \
      pf##SET = *reinterpret_cast<const float4*>(patch + pidx_ * C + 4 * ((2 * (Q) + fh) ^ ((pidx_ >> 1) & 7))); \
      wf##SET = *reinterpret_cast<const float4*>(w_lds + (TAP) * C * C + w_off[Q]);                    \
    }
#define UTP_MFMA(SET)                                                                                \
    {                                                                                                  \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf##SET.x, pf##SET.x, acc, 0, 0, 0);                  \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf##SET.y, pf##SET.y, acc, 0, 0, 0);                  \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf##SET.z, pf##SET.z, acc, 0, 0, 0);                  \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf##SET.w, pf##SET.w, acc, 0, 0, 0);                  \
    }
#define UTP_PIN() __builtin_amdgcn_sched_barrier(0)
      UTP_READ(X, 0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        UTP_READ(Y, tap, 1); UTP_PIN(); UTP_MFMA(X); UTP_PIN();
        if (has_next && tap < MAXP) issue_piece(tap, n_row0, n_y0, n_x0, cur ^ 1);
        if (has_next && tap == MAXP) init_load(n_row0, n_y0, n_x0);
        UTP_READ(X, tap, 2); UTP_PIN(); UTP_MFMA(Y); UTP_PIN();
        UTP_READ(Y, tap, 3); UTP_PIN(); UTP_MFMA(X); UTP_PIN();
        if (tap < 8) {
          UTP_READ(X, tap + 1, 0);
        } else {
          // everything requested for the next tile was issued at least two taps ago: drain the counter, publish
          // the ticket and combine bias+residual NOW, in front of the last MFMA group and of the stores (vmcnt
          // counts stores too: waiting at the barrier would cost every wave a write round trip per tile)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (tid == 0) slot_write(cur, grid + ticket);
          if (has_next) ready = init_combine();
          asm volatile("" : "+v"(ready));
        }
        UTP_PIN(); UTP_MFMA(Y); UTP_PIN();
      }
#undef UTP_READ
#undef UTP_MFMA
#undef UTP_PIN
    } else {
      // 9 taps x 2 k-steps of 16: per step two b128 reads of the lane's 8 patch values (already split: convert_patch), the
      // two weight planes (one conflict-free b128 each) and three MFMAs; the next step's reads are issued ahead of the MFMAs
      u32x4 prX[2], prY[2];      // the lane's 8 values of the step: first pieces, remainders
      u32x4 wqX[2], wqY[2];
      const char* w_bytes = reinterpret_cast<const char*>(smem) + lane * 16;
      const char* p_bytes = reinterpret_cast<const char*>(patch);
#define UTS_READ(SET, TAP, S)                                                                        \
  {                                                                                                  \
    const int pidx_ = pbase + ((TAP) / 3) * PW + ((TAP) % 3);                                        \
    const int off_ = pidx_ * 128 + (((2 * (S) + fh) ^ ((pidx_ >> 1) & 7)) << 4);                     \
    pr##SET[0] = *reinterpret_cast<const u32x4*>(p_bytes + off_);                                    \
    pr##SET[1] = *reinterpret_cast<const u32x4*>(p_bytes + (off_ ^ 64));                             \
    wq##SET[0] = *reinterpret_cast<const u32x4*>(w_bytes + (((TAP) * 2 + (S)) * 2 + 0) * 1024);      \
    wq##SET[1] = *reinterpret_cast<const u32x4*>(w_bytes + (((TAP) * 2 + (S)) * 2 + 1) * 1024);      \
  }
#define UTS_MFMA(SET)                                                                                \
  {                                                                                                  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8p, wq##SET[0]), __builtin_bit_cast(f16x8p, pr##SET[1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8p, wq##SET[1]), __builtin_bit_cast(f16x8p, pr##SET[0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8p, wq##SET[0]), __builtin_bit_cast(f16x8p, pr##SET[0]), acc, 0, 0, 0); \
  }
#define UTS_PIN() __builtin_amdgcn_sched_barrier(0)
      UTS_READ(X, 0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        UTS_READ(Y, tap, 1); UTS_PIN(); UTS_MFMA(X); UTS_PIN();
        if (has_next && tap < MAXP) issue_piece(tap, n_row0, n_y0, n_x0, cur ^ 1);
        if (has_next && tap == MAXP) init_load(n_row0, n_y0, n_x0);
        if (tap < 8) {
          UTS_READ(X, tap + 1, 0);
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (tid == 0) slot_write(cur, grid + ticket);
          if (has_next) ready = init_combine();
          asm volatile("" : "+v"(ready));
        }
        UTS_PIN(); UTS_MFMA(Y); UTS_PIN();
      }
#undef UTS_READ
#undef UTS_MFMA
#undef UTS_PIN
    }
    // epilogue: (ReLU) + 4 x 16-byte stores per lane.  Waves 0-3 store right after their MFMAs; waves 4-7 (the
    // SIMD partners of 0-3: a workgroup's waves w and w+4 share a SIMD) keep the tile in registers and store it at
    // the top of the NEXT tile instead, so that within a barrier interval one partner stores while the other
    // still feeds the matrix pipe - partners running the same program otherwise reach their store phase together.
    if (!late) { UTP_EPILOGUE((c_row0 + c_y0 + ly) * W + c_x0 + lx); }
    else { prev_m = (c_row0 + c_y0 + ly) * W + c_x0 + lx; have_prev = true; }
    if (!has_next) break;
    // every wave's pieces of the next patch have landed and everyone is done reading the current one
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
    if constexpr (SPLIT) {                // the next tile's patch is complete: split it in place, once
      convert_patch(cur ^ 1);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
    const int next2 = slot_read(cur);
    tile = next;
    next = next2;
    c_row0 = n_row0; c_y0 = n_y0; c_x0 = n_x0;
    cur ^= 1;
  }
  if (late && have_prev) { UTP_EPILOGUE(prev_m); }
  if constexpr (SPLIT) {
    if (p.out_max) publish_abs_max(p.out_max, out_bits);
  }
#undef UTP_EPILOGUE
}

bool conv_patch_applicable(const ConvLaunch& c) {
  return c.ksize == 3 && c.stride == 1 && c.pad == 1 && c.cin == C && c.cout_store == C && c.cslice == C &&
         c.k_pad == 9 * C && c.H % TY == 0 && c.W % TX == 0 && !c.out_nchw && c.H == c.Ho && c.W == c.Wo &&
         (size_t)c.n_img * c.H * c.W * C * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_patch(const ConvLaunch& c, hipStream_t s) {
  if (!conv_patch_applicable(c) || !c.tile_counter || c.num_cu <= 0) return hipErrorInvalidValue;
  const int tiles_x = c.W / TX, tiles_per_img = tiles_x * (c.H / TY);
  const int n_tiles = c.n_img * tiles_per_img;
  const size_t lds = (size_t)(W_FLOATS + 2 * P_FLOATS) * sizeof(float) + 16;
  // the attribute belongs to (kernel, device): one bit per device, set on the first launch there
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c32_patch_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c32_patch_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;           // one 512-thread workgroup per CU (LDS: 121 KB)
  if (grid > n_tiles) grid = n_tiles;
  if (c.w_split && c.split_unscale > 0.f)      // split-fp16 arithmetic (ut_set_conv_arithmetic)
    hipLaunchKernelGGL(conv3x3_c32_patch_kernel<true>, dim3(grid), dim3(64 * NW), lds, s, c, tiles_x, tiles_per_img, n_tiles);
  else
    hipLaunchKernelGGL(conv3x3_c32_patch_kernel<false>, dim3(grid), dim3(64 * NW), lds, s, c, tiles_x, tiles_per_img, n_tiles);
  return hipGetLastError();
}

}  // namespace ut
