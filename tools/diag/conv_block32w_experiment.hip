// The fused layer1 BasicBlock of conv_block32.hip with ONE wave per SIMD (4 waves, up to 512 registers each):
//     y = relu(bn2(conv2(relu(bn1(conv1 x)))) + x)          32 -> 32 -> 32 channels, 3x3, stride 1, split-fp16 arithmetic
// (lib/models/backbone_resnet.py:56-72 at 48x48x32).  The 8-wave kernel is bound by LDS bandwidth: with 32 channels per tap
// a (32 pixel x 32 channel x 32 k) step is 6 MFMAs on 4 KB of LDS reads, half of them the weight fragments that every
// wave and every tile re-reads (83 % LDS utilisation under conv1).  Here
//   * the 36 KB of conv1 weight fragments live in REGISTERS (144 per lane) for the life of the persistent workgroup, and
//     conv2's fragments, read from LDS, are shared by the two pixel blocks a wave computes: LDS reads per MFMA 0.67 -> 0.33 / 0.5;
//   * LDS without conv1's weights holds 16x16-pixel OUTPUT tiles: input patch 20x20, intermediate 18x18 = 324 pixels = 11 MFMA
//     blocks of 32 (three per wave; the twelfth is a dummy), conv2 256 pixels = 8 blocks (two per wave): matrix work per
//     output pixel 19/16 of the two-launch form instead of 14/12, and both convolutions balanced over the four SIMDs.
// Everything else as in conv_block32.hip: patch by LDS-DMA (double buffered across tiles), residual read from the fp32 patch
// before it is split in place, the intermediate split and written over the patch, its power-of-two scale from a bound.
#include <atomic>

#include "ut_kernels.h"

namespace ut {
bool conv_block32w_applicable(const BlockLaunch& b);
hipError_t launch_conv_block32w(const BlockLaunch& b, hipStream_t s);
namespace {

typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_charw;

constexpr int WT = 16;                               // output tile 16 x 16
constexpr int WI = WT + 2, WP = WT + 4;              // intermediate 18 x 18, patch 20 x 20
constexpr int WI_PIX = WI * WI, WP_PIX = WP * WP;    // 324, 400
constexpr int W_WAVES = 4;
constexpr int WU1 = 3, WU2 = 2;                      // pixel blocks per wave: conv1 (3 x 4 = 12 >= 11), conv2 (2 x 4 = 8)
static_assert(WU1 * W_WAVES * 32 >= WI_PIX && WU2 * W_WAVES * 32 == WT * WT, "blocks cover the tiles");
constexpr int W_PIECES = WP_PIX / 8;                 // 50 one-KB DMA pieces per patch
constexpr int W_MAXP = (W_PIECES + W_WAVES - 1) / W_WAVES;     // 13 per wave
constexpr int W_W_BYTES = 9 * 2 * 2 * 1024;          // conv2's planes in LDS: [tap][k-step][plane][lane][8 halves]
constexpr int W_R_BYTES = WP_PIX * 128;              // one patch / intermediate region: 51,200 B
constexpr int W_LDS = W_W_BYTES + 2 * W_R_BYTES + 64;
constexpr int WP_DIV = (65536 + WP - 1) / WP;        // r / 20 for r < 400
constexpr int WI_DIV = (65536 + WI - 1) / WI;        // q / 18 for q < 384
constexpr unsigned W_OOB = 0xFFFFFF00u;

__device__ __forceinline__ void w_dma(u32x4w rsrc, unsigned lds_addr, unsigned voffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc)
      : "memory");
}
__device__ __forceinline__ u32x4w w_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4w r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void w_split(float a, float b, unsigned& p0, unsigned& p1) {
  const f16x2w h = __builtin_bit_cast(f16x2w, __builtin_amdgcn_cvt_pkrtz(a, b));
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
__device__ __forceinline__ void w_split_scaled(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2w h = __builtin_bit_cast(f16x2w, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
__device__ __forceinline__ void w_pow2_for(float mx, float& scale, float& unscale) {
  const unsigned bits = __float_as_uint(mx);
  const int e = (int)(bits >> 23) & 0xFF;
  int k = (bits << 1) == 0u || e == 255 ? 0 : 141 - e;
  k = k > 100 ? 100 : k < -100 ? -100 : k;
  scale = __uint_as_float((unsigned)(127 + k) << 23);
  unscale = __uint_as_float((unsigned)(127 - k) << 23);
}
__device__ __forceinline__ f16x8w w_frag(u32x4w v) { return __builtin_bit_cast(f16x8w, v); }

}  // namespace

__global__ __launch_bounds__(64 * W_WAVES, 1) void conv_block32w_kernel(BlockLaunch p, int tiles_x, int tiles_per_img, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_charw*)smem;
  constexpr int W2_OFF = 0, R_OFF = W_W_BYTES, SLOT_OFF = R_OFF + 2 * W_R_BYTES;
  auto slot_write = [&](int idx, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(smem_addr + (unsigned)(SLOT_OFF + 4 * idx)), "v"(v) : "memory");
  };
  auto slot_read = [&](int idx) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(smem_addr + (unsigned)(SLOT_OFF + 4 * idx)) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int H = p.H, W = p.W;
  const int M = p.n_img * H * W;
  constexpr int C = 32;

  float x_scale = 1.f, x_unscale = 1.f, i_scale = 1.f, i_unscale = 1.f;
  {
    bool ok;
    split_act_scale(p.in_max, x_scale, x_unscale, ok);
#ifndef W4_STAMPS
    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);
#endif
    const float xmax = ok ? __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)*p.in_max)) : 0.f;
    w_pow2_for(xmax * p.wsum1 + p.bmax1, i_scale, i_unscale);
  }
  const float acc1_scale = x_scale / p.unscale_w1, acc1_unscale = p.unscale_w1 * x_unscale;
  const float acc2_scale = i_scale / p.unscale_w2, acc2_unscale = p.unscale_w2 * i_unscale;

  const u32x4w in_words = w_rsrc(p.in, (unsigned)((size_t)M * C * sizeof(float)));
  const u32x4w w2_words = w_rsrc(p.w2_split, (unsigned)W_W_BYTES);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)M * C * sizeof(float)), 0x00020000);

  // ---- conv2's planes -> LDS once (36 one-KB blocks); conv1's -> registers: [tap][k-step][plane], 16 bytes per lane each
  for (int k = wave; k < 36; k += W_WAVES) w_dma(w2_words, smem_addr + (unsigned)(W2_OFF + k * 1024), (unsigned)(k * 1024 + lane * 16));
  // (loaded straight into the accumulator half of the register file: a value that is born there stays there; loaded into a
  // vector register first, the compiler keeps that as its home and copies it over in front of every MFMA)
  u32x4w w1r[9][2][2];
  {
    const char* w1g = reinterpret_cast<const char*>(p.w1_split) + lane * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      asm volatile(
          "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
          "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&a"(w1r[t][0][0]), "=&a"(w1r[t][0][1]), "=&a"(w1r[t][1][0]), "=&a"(w1r[t][1][1])
          : "v"(w1g + t * 4096)
          : "memory");
  }

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
    y0 = ty * WT;
    x0 = c * WT;
  };
  auto issue_piece = [&](int j, int row0, int y0, int x0, int buf) {
    const int k = wave + W_WAVES * j;
    if (k < W_PIECES) {            // wave-uniform
      // which patch pixel / channel chunk this lane's 16 bytes of the piece are (recomputed: registers are for the weights)
      const int pidx = k * 8 + (lane >> 3);
      const int py = (pidx * WP_DIV) >> 16, px = pidx - py * WP;
      const int c4 = (lane & 7) ^ ((pidx >> 1) & 7);
      const int gy = y0 - 2 + py, gx = x0 - 2 + px;
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)((((row0 + gy) * W + gx) * C + 4 * c4) * 4) : W_OOB;
      w_dma(in_words, (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_addr + (unsigned)(R_OFF + buf * W_R_BYTES + k * 1024))), off);
    }
  };

  // ---- my pixels: conv1 block j of this wave = intermediate pixels 32 * (wave + 4 j) + fr of the 18x18 raster (the twelfth
  // block and the last 28 lanes of the eleventh are dummies: they read a valid row, their results are not written);
  // conv2 block j = output pixels 32 * (wave + 4 j) + fr of the 16x16 raster
  int q1[WU1], p1base[WU1], iy1[WU1], ix1[WU1];
#pragma unroll
  for (int j = 0; j < WU1; ++j) {
    q1[j] = 32 * (wave + W_WAVES * j) + fr;
    const int qc = q1[j] < WI_PIX ? q1[j] : WI_PIX - 1;
    iy1[j] = (qc * WI_DIV) >> 16;
    ix1[j] = qc - iy1[j] * WI;
    p1base[j] = iy1[j] * WP + ix1[j];
  }
  int oy2[WU2], ox2[WU2], p2base[WU2], res_row[WU2];
#pragma unroll
  for (int j = 0; j < WU2; ++j) {
    const int q2 = 32 * (wave + W_WAVES * j) + fr;
    oy2[j] = q2 >> 4;
    ox2[j] = q2 & 15;
    p2base[j] = oy2[j] * WI + ox2[j];
    res_row[j] = (oy2[j] + 2) * WP + ox2[j] + 2;
  }

  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.tile_counter, 0, 4, 0x00020000);
  const unsigned q_off = tid == 0 ? 0u : W_OOB;

  const int grid = gridDim.x;
  int tile = blockIdx.x;
  int c_row0, c_y0, c_x0;
  tile_origin(tile, c_row0, c_y0, c_x0);
#pragma unroll
  for (int j = 0; j < W_MAXP; ++j) issue_piece(j, c_row0, c_y0, c_x0, 0);
  {
    const int t0 = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);
    if (tid == 0) slot_write(2, grid + t0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  int next = slot_read(2);
  int cur = 0;
  unsigned out_bits = 0;
  const char* w2_bytes = smem + W2_OFF + lane * 16;

  auto convert_row = [&](char* reg, int row) {
    const int sw = (row >> 1) & 7;
    char* rp = reg + row * 128;
    float4 f[8];
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4) f[g4] = *reinterpret_cast<const float4*>(rp + ((g4 ^ sw) << 4));
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      unsigned a0, a1, a2, a3, b0, b1, b2, b3;
      w_split_scaled(f[2 * kg].x, f[2 * kg].y, x_scale, a0, b0);
      w_split_scaled(f[2 * kg].z, f[2 * kg].w, x_scale, a1, b1);
      w_split_scaled(f[2 * kg + 1].x, f[2 * kg + 1].y, x_scale, a2, b2);
      w_split_scaled(f[2 * kg + 1].z, f[2 * kg + 1].w, x_scale, a3, b3);
      u32x4w a, b;
      a.x = a0; a.y = a1; a.z = a2; a.w = a3;
      b.x = b0; b.y = b1; b.z = b2; b.w = b3;
      *reinterpret_cast<u32x4w*>(rp + ((kg ^ sw) << 4)) = a;
      *reinterpret_cast<u32x4w*>(rp + (((4 + kg) ^ sw) << 4)) = b;
    }
  };

#define W_PIN() __builtin_amdgcn_sched_barrier(0)
  // the lane's 8 values (both pieces) of block J for tap TAP, k-step S, out of region REG with row width ROWW
#define W_READ_PX(DST, REG, PBASE, ROWW, TAP, S)                                                     \
  {                                                                                                  \
    const int pidx_ = (PBASE) + ((TAP) / 3) * (ROWW) + ((TAP) % 3);                                  \
    const int off_ = pidx_ * 128 + (((2 * (S) + fh) ^ ((pidx_ >> 1) & 7)) << 4);                     \
    DST[0] = *reinterpret_cast<const u32x4w*>((REG) + off_);                                         \
    DST[1] = *reinterpret_cast<const u32x4w*>((REG) + (off_ ^ 64));                                  \
  }
  // The MFMAs are inline asm so that the operand classes are mine: accumulators AND conv1's 144 weight registers in the
  // accumulator half of the register file ("a"), pixel fragments in the vector half ("v") - the compiler's own choice keeps
  // MFMA A/B operands in the vector half, which then spills.  What the compiler does not know about an asm MFMA: its result
  // needs 12+ wait states before anything but the next MFMA of the chain reads it (W_MFMA_DRAIN behind each loop), and an
  // operand written by a vector instruction just before it needs 2 (W_MFMA_LEAD in front of each loop).
#define W_MFMA1(ACC, WCL, WV, PXV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : WCL(WV), "v"(PXV))
#define W_MFMA3(ACC, WCL, W0, W1, PX)                                                                \
  {                                                                                                  \
    W_MFMA1(ACC, WCL, W0, PX[1]);                                                                    \
    W_MFMA1(ACC, WCL, W1, PX[0]);                                                                    \
    W_MFMA1(ACC, WCL, W0, PX[0]);                                                                    \
  }
#define W_MFMA_LEAD() asm volatile("s_nop 3")
#define W_MFMA_DRAIN3(A, B, C) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(A), "+a"(B), "+a"(C))
#define W_MFMA_DRAIN2(A, B) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(A), "+a"(B))

#ifdef W4_STAMPS
  int tile_no = 0;
  unsigned long long st_[8];
#define W_STAMP(I) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[I]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define W_STAMP(I)
#endif
  for (;;) {
    const bool has_next = (unsigned)next < (unsigned)n_tiles;
    char* region = smem + R_OFF + cur * W_R_BYTES;
    W_STAMP(0)
    // per-tile opaque copies of the row bases: the per-tap address arithmetic then stays inside the tile loop instead of being
    // hoisted into ~100 registers that live across it
    int p1b[WU1], p2b[WU2];
#pragma unroll
    for (int j = 0; j < WU1; ++j) { p1b[j] = p1base[j]; asm volatile("" : "+v"(p1b[j])); }
#pragma unroll
    for (int j = 0; j < WU2; ++j) { p2b[j] = p2base[j]; asm volatile("" : "+v"(p2b[j])); }
    // ---- A: residuals of my two output blocks from the fp32 patch
    u32x4w rr[WU2][4];
#pragma unroll
    for (int j = 0; j < WU2; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        rr[j][g4] = *reinterpret_cast<const u32x4w*>(region + res_row[j] * 128 + (((2 * g4 + fh) ^ ((res_row[j] >> 1) & 7)) << 4));
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S1
    // ---- B: split the patch in place
    convert_row(region, tid);
    if (tid + 256 < WP_PIX) convert_row(region, tid + 256);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S2
    W_STAMP(1)

    // ---- C: conv1 on my three blocks, weights from registers; the next patch is requested piece by piece under it
    const int ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);
    int n_row0 = 0, n_y0 = 0, n_x0 = 0;
    if (has_next) tile_origin(next, n_row0, n_y0, n_x0);
    f32x16w acc[WU1];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias1 + 8 * g4 + 4 * fh);      // (L1-resident; not kept in registers)
#pragma unroll
      for (int j = 0; j < WU1; ++j) {
        acc[j][4 * g4 + 0] = b1.x * acc1_scale; acc[j][4 * g4 + 1] = b1.y * acc1_scale;
        acc[j][4 * g4 + 2] = b1.z * acc1_scale; acc[j][4 * g4 + 3] = b1.w * acc1_scale;
      }
    }
    {
      u32x4w pxA[WU1][2], pxB[WU1][2];
#pragma unroll
      for (int j = 0; j < WU1; ++j) W_READ_PX(pxA[j], region, p1b[j], WP, 0, 0);
#pragma unroll
      for (int j = 0; j < WU1; ++j) asm volatile("" : "+a"(acc[j]));      // initialised before the first MFMA, not between them
      W_MFMA_LEAD();
      // One wave per SIMD issues in order: whatever is not an MFMA must sit BETWEEN the MFMAs (each leaves ~24 idle issue
      // cycles), not between groups of them.  Block j's fragments of the NEXT k-step are requested right behind block j's three
      // MFMAs of this one; a patch piece of the next tile rides behind the first block.
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        W_MFMA3(acc[0], "a", w1r[tap][0][0], w1r[tap][0][1], pxA[0]); W_PIN();
        W_READ_PX(pxB[0], region, p1b[0], WP, tap, 1);
        if (has_next && 2 * tap < W_MAXP) issue_piece(2 * tap, n_row0, n_y0, n_x0, cur ^ 1);
        W_PIN();
        W_MFMA3(acc[1], "a", w1r[tap][0][0], w1r[tap][0][1], pxA[1]); W_PIN();
        W_READ_PX(pxB[1], region, p1b[1], WP, tap, 1); W_PIN();
        W_MFMA3(acc[2], "a", w1r[tap][0][0], w1r[tap][0][1], pxA[2]); W_PIN();
        W_READ_PX(pxB[2], region, p1b[2], WP, tap, 1); W_PIN();
        W_MFMA3(acc[0], "a", w1r[tap][1][0], w1r[tap][1][1], pxB[0]); W_PIN();
        if (tap < 8) W_READ_PX(pxA[0], region, p1b[0], WP, tap + 1, 0);
        if (has_next && 2 * tap + 1 < W_MAXP) issue_piece(2 * tap + 1, n_row0, n_y0, n_x0, cur ^ 1);
        W_PIN();
        W_MFMA3(acc[1], "a", w1r[tap][1][0], w1r[tap][1][1], pxB[1]); W_PIN();
        if (tap < 8) W_READ_PX(pxA[1], region, p1b[1], WP, tap + 1, 0);
        W_PIN();
        W_MFMA3(acc[2], "a", w1r[tap][1][0], w1r[tap][1][1], pxB[2]); W_PIN();
        if (tap < 8) W_READ_PX(pxA[2], region, p1b[2], WP, tap + 1, 0);
        W_PIN();
      }
    }
    W_STAMP(2)
    W_MFMA_DRAIN3(acc[0], acc[1], acc[2]);
    // BatchNorm (folded) + ReLU; intermediate pixels outside the image are conv2's zero padding
    unsigned ip0[WU1][8], ip1[WU1][8];
#pragma unroll
    for (int j = 0; j < WU1; ++j) {
      const int gy = c_y0 - 1 + iy1[j], gx = c_x0 - 1 + ix1[j];
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float keep = inside ? i_scale : 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float v0 = fmaxf(acc[j][4 * g4 + 0] * acc1_unscale, 0.f) * keep, v1 = fmaxf(acc[j][4 * g4 + 1] * acc1_unscale, 0.f) * keep;
        const float v2 = fmaxf(acc[j][4 * g4 + 2] * acc1_unscale, 0.f) * keep, v3 = fmaxf(acc[j][4 * g4 + 3] * acc1_unscale, 0.f) * keep;
        w_split(v0, v1, ip0[j][2 * g4], ip1[j][2 * g4]);
        w_split(v2, v3, ip0[j][2 * g4 + 1], ip1[j][2 * g4 + 1]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S3: nobody reads the patch any more
#pragma unroll
    for (int j = 0; j < WU1; ++j) {
      if (q1[j] < WI_PIX) {
        const int sw = (q1[j] >> 1) & 7;
        char* rp = region + q1[j] * 128 + 8 * fh;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          u32x2w a, b;
          a.x = ip0[j][2 * g4]; a.y = ip0[j][2 * g4 + 1];
          b.x = ip1[j][2 * g4]; b.y = ip1[j][2 * g4 + 1];
          *reinterpret_cast<u32x2w*>(rp + ((g4 ^ sw) << 4)) = a;
          *reinterpret_cast<u32x2w*>(rp + (((4 + g4) ^ sw) << 4)) = b;
        }
      }
    }
    if (tid == 0) slot_write(cur, grid + ticket);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S4: the intermediate is complete
    W_STAMP(3)

    // ---- D: conv2 on my two blocks; one weight-fragment read from LDS serves both
    f32x16w acc2[WU2];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 b2 = *reinterpret_cast<const float4*>(p.bias2 + 8 * g4 + 4 * fh);
#pragma unroll
      for (int j = 0; j < WU2; ++j) {
        acc2[j][4 * g4 + 0] = (b2.x + __uint_as_float(rr[j][g4].x)) * acc2_scale;
        acc2[j][4 * g4 + 1] = (b2.y + __uint_as_float(rr[j][g4].y)) * acc2_scale;
        acc2[j][4 * g4 + 2] = (b2.z + __uint_as_float(rr[j][g4].z)) * acc2_scale;
        acc2[j][4 * g4 + 3] = (b2.w + __uint_as_float(rr[j][g4].w)) * acc2_scale;
      }
    }
    {
      u32x4w pxA[WU2][2], pxB[WU2][2], wA[2], wB[2];
#define W_READ_W(DST, TAP, S)                                                                        \
  {                                                                                                  \
    DST[0] = *reinterpret_cast<const u32x4w*>(w2_bytes + (((TAP) * 2 + (S)) * 2 + 0) * 1024);        \
    DST[1] = *reinterpret_cast<const u32x4w*>(w2_bytes + (((TAP) * 2 + (S)) * 2 + 1) * 1024);        \
  }
#pragma unroll
      for (int j = 0; j < WU2; ++j) W_READ_PX(pxA[j], region, p2b[j], WI, 0, 0);
      W_READ_W(wA, 0, 0);
#pragma unroll
      for (int j = 0; j < WU2; ++j) asm volatile("" : "+a"(acc2[j]));
      W_MFMA_LEAD();
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        W_MFMA3(acc2[0], "v", wA[0], wA[1], pxA[0]); W_PIN();
        W_READ_PX(pxB[0], region, p2b[0], WI, tap, 1);
        W_READ_W(wB, tap, 1);
        W_PIN();
        W_MFMA3(acc2[1], "v", wA[0], wA[1], pxA[1]); W_PIN();
        W_READ_PX(pxB[1], region, p2b[1], WI, tap, 1);
        if (tap < 8) W_READ_W(wA, tap + 1, 0);          // wA is free from here: three groups ahead of its next use
        W_PIN();
        W_MFMA3(acc2[0], "v", wB[0], wB[1], pxB[0]); W_PIN();
        if (tap < 8) {
          W_READ_PX(pxA[0], region, p2b[0], WI, tap + 1, 0);
        } else {
          // the next patch has had conv1 and conv2 to land; nothing else of mine is in flight, and this sits in front of
          // the tile's stores (vmcnt counts stores too: behind them the wait would cost a write round trip)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        W_PIN();
        W_MFMA3(acc2[1], "v", wB[0], wB[1], pxB[1]); W_PIN();
        if (tap < 8) W_READ_PX(pxA[1], region, p2b[1], WI, tap + 1, 0);
        W_PIN();
      }
#undef W_READ_W
    }
    W_STAMP(4)
    W_MFMA_DRAIN2(acc2[0], acc2[1]);
#pragma unroll
    for (int j = 0; j < WU2; ++j) {
      const int m = (c_row0 + c_y0 + oy2[j]) * W + c_x0 + ox2[j];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x4w pk;
        pk.x = __float_as_uint(fmaxf(acc2[j][4 * g4 + 0] * acc2_unscale, 0.f));
        pk.y = __float_as_uint(fmaxf(acc2[j][4 * g4 + 1] * acc2_unscale, 0.f));
        pk.z = __float_as_uint(fmaxf(acc2[j][4 * g4 + 2] * acc2_unscale, 0.f));
        pk.w = __float_as_uint(fmaxf(acc2[j][4 * g4 + 3] * acc2_unscale, 0.f));
        out_bits = max(max(out_bits, max(pk.x & 0x7FFFFFFFu, pk.y & 0x7FFFFFFFu)), max(pk.z & 0x7FFFFFFFu, pk.w & 0x7FFFFFFFu));
        __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, (unsigned)(m * C + 8 * g4 + 4 * fh) * 4u, 0, 0);
      }
    }
#ifdef W4_STAMPS
    W_STAMP(5)
    if (tile_no == 3 && tid == 0) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(p.status) + blockIdx.x * 8;
      for (int i = 0; i < 6; ++i) d[i] = st_[i];
    }
    ++tile_no;
#endif
    if (!has_next) break;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S0: the next patch has landed for every wave; conv2's reads of this region are done
    const int next2 = slot_read(cur);        // written before S4
    tile = next;
    next = next2;
    c_row0 = n_row0; c_y0 = n_y0; c_x0 = n_x0;
    cur ^= 1;
  }
  if (p.out_max) publish_abs_max(p.out_max, out_bits);
#undef W_PIN
#undef W_READ_PX
#undef W_MFMA3
#undef W_MFMA1
#undef W_MFMA_LEAD
#undef W_MFMA_DRAIN3
#undef W_MFMA_DRAIN2
}

bool conv_block32w_applicable(const BlockLaunch& b) {
  return b.in && b.out && b.w1_split && b.w2_split && b.bias1 && b.bias2 && b.in_max && b.tile_counter && b.num_cu > 0 &&
         b.unscale_w1 > 0.f && b.unscale_w2 > 0.f && b.n_img > 0 && b.H % WT == 0 && b.W % WT == 0 &&
         (size_t)b.n_img * b.H * b.W * 32 * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_block32w(const BlockLaunch& b, hipStream_t s) {
  if (!conv_block32w_applicable(b)) return hipErrorInvalidValue;
  const int tiles_x = b.W / WT, tiles_per_img = tiles_x * (b.H / WT);
  const int n_tiles = b.n_img * tiles_per_img;
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (b.device >= 0 && b.device < 64) ? 1ull << b.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_block32w_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = b.num_cu;           // one 256-thread workgroup per CU (LDS: 139 KB; one wave per SIMD)
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(conv_block32w_kernel, dim3(grid), dim3(64 * W_WAVES), W_LDS, s, b, tiles_x, tiles_per_img, n_tiles);
  return hipGetLastError();
}

}  // namespace ut
