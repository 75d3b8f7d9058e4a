// One ResNet BasicBlock of layer1 in ONE launch (split-fp16 arithmetic):
//     y = relu(bn2(conv2(relu(bn1(conv1 x)))) + x)          32 -> 32 -> 32 channels, 3x3, stride 1
// (lib/models/backbone_resnet.py:56-72 at 48x48x32).  As two launches (conv_patch.hip<SPLIT>) each convolution is bound
// by its tile's HBM traffic - patch in, residual in, tile out: 3.8 TB/s at 1 ms per launch - and the intermediate
// relu(bn1(conv1 x)) makes a full round trip through HBM.  Here a workgroup owns a 12x16-pixel OUTPUT tile:
//   * the 16x20x32 input patch (tile + a halo of 2) comes into LDS once by LDS-DMA, double buffered across tiles;
//   * the patch is split IN PLACE into the two fp16 pieces per value (a value's pieces take its 4 bytes; scaled by the power
//     of two of ut_kernels.h::split_act_scale) - by waves 6 and 7, which have no conv2 block, while waves 0..5 run conv2 of
//     the tile before; the residual (the fp32 centre pixels) is read into registers by waves 0..5 just before that;
//   * conv1 runs on the 14x18 intermediate pixels (tile + halo of 1: 252 pixels = 8 MFMA blocks of 32, one per wave),
//     BatchNorm + ReLU applied, intermediate pixels outside the image set to zero (they are conv2's zero padding);
//   * the intermediate never leaves the CU: once every wave is done reading the patch it is split and written over the patch;
//   * conv2 reads it there (192 pixels = 6 MFMA blocks: waves 0..5), adds bias + residual, ReLU, 16-byte NHWC stores.
// HBM traffic per block: the input once (+ the halo overlap, mostly L2 hits) and the output once - 2 x 1.2 GB per 4096 crops
// instead of 5 x 1.2 GB; matrix work +17 % (conv1 also computes the halo ring).  Both weight tensors (2 x 36 KB of fp16
// planes in fragment order) stay resident in LDS for the life of the persistent workgroup: 155.8 of the CU's 160 KB.
//
// Activation scale of the intermediate.  The split needs a power of two that brings the intermediate under 2^15 BEFORE any
// of it exists, so it cannot be the producer's max word.  It is a bound instead: |relu(bn1(conv1 x))| <= max|x| * max_c sum_k
// |w1[c][k]| + max|b1| (host: the row sums of the folded weights; device: max|x| from the input's max word).  The bound is
// loose by the usual gap between an L1 and a random-sign sum (tens), which costs nothing: a value keeps its full 22 bits
// down to 2^-18 of the bound and an absolute 2^-40 of the bound below.  Same bits for every tiling and batch.
//
// MFMA operand roles, LDS row format and swizzles as in conv_patch.hip: weights = "A" (rows = output channel), pixels =
// "B" (columns): a lane owns one pixel, accumulator register quads are 4 consecutive channels.
#include <atomic>

#include "ut_kernels.h"

namespace ut {
namespace {

typedef float f32x16b __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8b __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2b __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_charb;

constexpr int BT_Y = 12, BT_X = 16;                 // output tile
constexpr int BI_W = BT_X + 2, BI_H = BT_Y + 2;     // intermediate (conv1 output) tile: 18 x 14 = 252 pixels
constexpr int BP_W = BT_X + 4, BP_H = BT_Y + 4;     // input patch: 20 x 16 = 320 pixels
constexpr int BI_PIX = BI_W * BI_H, BP_PIX = BP_W * BP_H;
constexpr int BO_PIX = BT_X * BT_Y;                 // 192
constexpr int B_WAVES = 8;
constexpr int U1 = (BI_PIX + 31) / 32, U2 = BO_PIX / 32;      // MFMA pixel blocks of conv1 (8) and conv2 (6)
static_assert(U1 == B_WAVES && U2 <= B_WAVES && BO_PIX % 32 == 0, "one conv1 block per wave");
static_assert(BP_PIX % 8 == 0, "whole 8-row DMA pieces");
constexpr int B_PIECES = BP_PIX / 8;                // 40 one-KB pieces per patch
constexpr int B_MAXP = (B_PIECES + B_WAVES - 1) / B_WAVES;    // 5 per wave
constexpr int B_W_BYTES = 9 * 2 * 2 * 1024;         // one convolution's planes: [tap][k-step][plane][lane][8 halves]
constexpr int B_R_BYTES = BP_PIX * 128;             // one patch / intermediate region
constexpr int B_LDS = 2 * B_W_BYTES + 2 * B_R_BYTES + 64;
constexpr int BP_DIV = (65536 + BP_W - 1) / BP_W;   // r / 20 == (r * BP_DIV) >> 16 for r < 320
constexpr int BI_DIV = (65536 + BI_W - 1) / BI_W;   // q / 18 == (q * BI_DIV) >> 16 for q < 256
constexpr unsigned B_OOB = 0xFFFFFF00u;

__device__ __forceinline__ void b_dma(u32x4b rsrc, unsigned lds_addr, unsigned voffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc)
      : "memory");
}
__device__ __forceinline__ u32x4b b_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4b r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void b_split(float a, float b, unsigned& p0, unsigned& p1) {
  const f16x2b h = __builtin_bit_cast(f16x2b, __builtin_amdgcn_cvt_pkrtz(a, b));
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
// the two pieces of a * s and b * s for a power of two s: the products are exact, so fma(a, s, -h) is the same remainder as
// (a * s) - h, in one instruction that also converts h (v_fma_mix_f32)
__device__ __forceinline__ void b_split_scaled(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2b h = __builtin_bit_cast(f16x2b, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
// 2^k with max * 2^k in [2^14, 2^15) for a positive finite `max` (1 for 0), and its inverse
__device__ __forceinline__ void b_pow2_for(float mx, float& scale, float& unscale) {
  const unsigned bits = __float_as_uint(mx);
  const int e = (int)(bits >> 23) & 0xFF;
  int k = (bits << 1) == 0u || e == 255 ? 0 : 141 - e;
  k = k > 100 ? 100 : k < -100 ? -100 : k;
  scale = __uint_as_float((unsigned)(127 + k) << 23);
  unscale = __uint_as_float((unsigned)(127 - k) << 23);
}

}  // namespace

__global__ __launch_bounds__(64 * B_WAVES) void conv_block32_kernel(BlockLaunch p, int tiles_x, int tiles_per_img, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_charb*)smem;
  constexpr int W1_OFF = 0, W2_OFF = B_W_BYTES, R_OFF = 2 * B_W_BYTES, SLOT_OFF = R_OFF + 2 * B_R_BYTES;
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

  // ---- scales: input (from the producer's max word), intermediate (a bound, see the header)
  float x_scale = 1.f, x_unscale = 1.f, i_scale = 1.f, i_unscale = 1.f;
  {
    bool ok;
    split_act_scale(p.in_max, p.in_obs, x_scale, x_unscale, ok);
    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);
    const float xmax = ok ? __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)*p.in_max)) : 0.f;
    b_pow2_for(xmax * p.wsum1 + p.bmax1, i_scale, i_unscale);
  }
  const float acc1_scale = x_scale / p.unscale_w1, acc1_unscale = p.unscale_w1 * x_unscale;     // powers of two
  const float acc2_scale = i_scale / p.unscale_w2, acc2_unscale = p.unscale_w2 * i_unscale;

  const u32x4b in_words = b_rsrc(p.in, (unsigned)((size_t)M * C * sizeof(float)));
  const u32x4b w1_words = b_rsrc(p.w1_split, (unsigned)B_W_BYTES), w2_words = b_rsrc(p.w2_split, (unsigned)B_W_BYTES);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)M * C * sizeof(float)), 0x00020000);

  // ---- both weight tensors -> LDS once: 2 x 36 one-KB blocks, rows 0..31 of ConvW::w_split (conv_split.hip::pack_split_weights)
  for (int k = wave; k < 72; k += B_WAVES) {
    if (k < 36) b_dma(w1_words, smem_addr + (unsigned)(W1_OFF + k * 1024), (unsigned)(k * 1024 + lane * 16));
    else b_dma(w2_words, smem_addr + (unsigned)(W2_OFF + (k - 36) * 1024), (unsigned)((k - 36) * 1024 + lane * 16));
  }

  // ---- per-lane constants of the patch DMA: which patch pixel / channel chunk each of my pieces is
  int pc_py[B_MAXP], pc_px[B_MAXP], pc_c4[B_MAXP];
#pragma unroll
  for (int j = 0; j < B_MAXP; ++j) {
    const int k = wave + B_WAVES * j;                // piece index (wave-uniform), always < B_PIECES (40 = 8 x 5)
    const int pidx = k * 8 + (lane >> 3), cpos = lane & 7;
    const int py = (pidx * BP_DIV) >> 16;
    pc_py[j] = py;
    pc_px[j] = pidx - py * BP_W;
    pc_c4[j] = cpos ^ ((pidx >> 1) & 7);
  }
  static_assert(B_PIECES == B_WAVES * B_MAXP, "every wave issues B_MAXP pieces");

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
    y0 = ty * BT_Y;
    x0 = c * BT_X;
  };
  auto issue_piece = [&](int j, int row0, int y0, int x0, int buf) {
    const int k = wave + B_WAVES * j;
    const int gy = y0 - 2 + pc_py[j], gx = x0 - 2 + pc_px[j];
    const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const unsigned off = ok ? (unsigned)((((row0 + gy) * W + gx) * C + 4 * pc_c4[j]) * 4) : B_OOB;
    b_dma(in_words, smem_addr + (unsigned)(R_OFF + buf * B_R_BYTES + k * 1024), off);
  };

  // ---- my pixels.  conv1: intermediate pixel q1 = 32 * wave + fr of the 14x18 raster (the last block has 4 spare lanes);
  // conv2 (waves 0..5): output pixel q2 = 32 * wave + fr of the 12x16 raster.
  const int q1 = 32 * wave + fr;
  const bool q1_real = q1 < BI_PIX;
  const int q1c = q1_real ? q1 : BI_PIX - 1;           // spare lanes read a valid row (their results are discarded)
  const int iy = (q1c * BI_DIV) >> 16, ix = q1c - iy * BI_W;
  const int p1base = iy * BP_W + ix;                   // patch row of tap (0, 0)
  const bool has_u2 = wave < U2;
  const int q2 = has_u2 ? 32 * wave + fr : fr;
  const int oy = q2 >> 4, ox = q2 & 15;
  static_assert(BT_X == 16, "q2 >> 4");
  const int p2base = oy * BI_W + ox;                   // intermediate row of tap (0, 0)

  float4 b1r[4], b2r[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    b1r[g4] = *reinterpret_cast<const float4*>(p.bias1 + 8 * g4 + 4 * fh);
    b2r[g4] = *reinterpret_cast<const float4*>(p.bias2 + 8 * g4 + 4 * fh);
  }

  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.tile_counter, 0, 4, 0x00020000);
  const unsigned q_off = tid == 0 ? 0u : B_OOB;

  const int grid = gridDim.x;
  int tile = blockIdx.x;
  int c_row0, c_y0, c_x0;
  tile_origin(tile, c_row0, c_y0, c_x0);
#pragma unroll
  for (int j = 0; j < B_MAXP; ++j) issue_piece(j, c_row0, c_y0, c_x0, 0);
  {
    const int t0 = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);
    if (tid == 0) slot_write(2, grid + t0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  int next = slot_read(2);
  int cur = 0;
  unsigned out_bits = 0;

  const char* w1_bytes = smem + W1_OFF + lane * 16;
  const char* w2_bytes = smem + W2_OFF + lane * 16;

  // split one patch row in place (group q = 4 * piece + k / 8 at position q ^ swizzle)
  auto convert_row = [&](char* reg, int row) {
    const int sw = (row >> 1) & 7;
    char* rp = reg + row * 128;
    float4 f[8];
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4) f[g4] = *reinterpret_cast<const float4*>(rp + ((g4 ^ sw) << 4));
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      unsigned a0, a1, a2, a3, b0, b1, b2, b3;
      b_split_scaled(f[2 * kg].x, f[2 * kg].y, x_scale, a0, b0);
      b_split_scaled(f[2 * kg].z, f[2 * kg].w, x_scale, a1, b1);
      b_split_scaled(f[2 * kg + 1].x, f[2 * kg + 1].y, x_scale, a2, b2);
      b_split_scaled(f[2 * kg + 1].z, f[2 * kg + 1].w, x_scale, a3, b3);
      u32x4b a, b;
      a.x = a0; a.y = a1; a.z = a2; a.w = a3;
      b.x = b0; b.y = b1; b.z = b2; b.w = b3;
      *reinterpret_cast<u32x4b*>(rp + ((kg ^ sw) << 4)) = a;
      *reinterpret_cast<u32x4b*>(rp + (((4 + kg) ^ sw) << 4)) = b;
    }
  };
  // the residual of my output pixel (waves 0..5): the fp32 centre of a landed patch, read before the patch is split
  const int res_row = (oy + 2) * BP_W + ox + 2;
  u32x4b rr[4];
  auto read_residual = [&](const char* reg) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      rr[g4] = *reinterpret_cast<const u32x4b*>(reg + res_row * 128 + (((2 * g4 + fh) ^ ((res_row >> 1) & 7)) << 4));
  };
  // the first tile's patch: residual, then everybody converts
  if (has_u2) read_residual(smem + R_OFF);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  if (tid < BP_PIX) convert_row(smem + R_OFF, tid);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();

#define B_PIN() __builtin_amdgcn_sched_barrier(0)
#define B_READ(SET, REGION, WB, PBASE, ROWW, TAP, S)                                                 \
  {                                                                                                  \
    const int pidx_ = (PBASE) + ((TAP) / 3) * (ROWW) + ((TAP) % 3);                                  \
    const int off_ = pidx_ * 128 + (((2 * (S) + fh) ^ ((pidx_ >> 1) & 7)) << 4);                     \
    pr##SET[0] = *reinterpret_cast<const u32x4b*>((REGION) + off_);                                  \
    pr##SET[1] = *reinterpret_cast<const u32x4b*>((REGION) + (off_ ^ 64));                           \
    wq##SET[0] = *reinterpret_cast<const u32x4b*>((WB) + (((TAP) * 2 + (S)) * 2 + 0) * 1024);        \
    wq##SET[1] = *reinterpret_cast<const u32x4b*>((WB) + (((TAP) * 2 + (S)) * 2 + 1) * 1024);        \
  }
#define B_MFMA(SET)                                                                                  \
  {                                                                                                  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8b, wq##SET[0]), __builtin_bit_cast(f16x8b, pr##SET[1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8b, wq##SET[1]), __builtin_bit_cast(f16x8b, pr##SET[0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8b, wq##SET[0]), __builtin_bit_cast(f16x8b, pr##SET[0]), acc, 0, 0, 0); \
  }

  for (;;) {
    const bool has_next = (unsigned)next < (unsigned)n_tiles;     // (unsigned: a corrupt queue word cannot keep the loop alive)
    char* region = smem + R_OFF + cur * B_R_BYTES;
    // ---- C: conv1 on my 32 intermediate pixels.  The next tile's patch pieces and the queue ticket are requested between
    // the MFMA groups of the first taps (the other region is free: its last readers, conv2 of the previous tile, passed S2).
    const int ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);
    int n_row0 = 0, n_y0 = 0, n_x0 = 0;
    if (has_next) tile_origin(next, n_row0, n_y0, n_x0);
    f32x16b acc;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      acc[4 * g4 + 0] = b1r[g4].x * acc1_scale; acc[4 * g4 + 1] = b1r[g4].y * acc1_scale;
      acc[4 * g4 + 2] = b1r[g4].z * acc1_scale; acc[4 * g4 + 3] = b1r[g4].w * acc1_scale;
    }
    {
      u32x4b prX[2], prY[2], wqX[2], wqY[2];
      B_READ(X, region, w1_bytes, p1base, BP_W, 0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        B_READ(Y, region, w1_bytes, p1base, BP_W, tap, 1); B_PIN(); B_MFMA(X); B_PIN();
        if (has_next && tap < B_MAXP) issue_piece(tap, n_row0, n_y0, n_x0, cur ^ 1);
        if (tap < 8) B_READ(X, region, w1_bytes, p1base, BP_W, tap + 1, 0);
        B_PIN(); B_MFMA(Y); B_PIN();
      }
    }
    // BatchNorm (folded) + ReLU; intermediate pixels outside the image are conv2's zero padding
    unsigned ip0[8], ip1[8];
    {
      const int gy = c_y0 - 1 + iy, gx = c_x0 - 1 + ix;
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float keep = inside ? i_scale : 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float v0 = fmaxf(acc[4 * g4 + 0] * acc1_unscale, 0.f) * keep, v1 = fmaxf(acc[4 * g4 + 1] * acc1_unscale, 0.f) * keep;
        const float v2 = fmaxf(acc[4 * g4 + 2] * acc1_unscale, 0.f) * keep, v3 = fmaxf(acc[4 * g4 + 3] * acc1_unscale, 0.f) * keep;
        b_split(v0, v1, ip0[2 * g4], ip1[2 * g4]);
        b_split(v2, v3, ip0[2 * g4 + 1], ip1[2 * g4 + 1]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S3: nobody reads the patch any more
    // ---- the intermediate over the patch: row q1, channels 8 g4 + 4 fh .. + 3 = half fh of group g4 (piece 0) / 4 + g4 (piece 1)
    if (q1_real) {
      const int sw = (q1 >> 1) & 7;
      char* rp = region + q1 * 128 + 8 * fh;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x2b a, b;
        a.x = ip0[2 * g4]; a.y = ip0[2 * g4 + 1];
        b.x = ip1[2 * g4]; b.y = ip1[2 * g4 + 1];
        *reinterpret_cast<u32x2b*>(rp + ((g4 ^ sw) << 4)) = a;
        *reinterpret_cast<u32x2b*>(rp + (((4 + g4) ^ sw) << 4)) = b;
      }
    }
    // the next tile's patch has had all of conv1 to land; nothing else of mine is in flight (the previous tile's stores are
    // long done), so this wait costs no store round trip.  Publish the ticket for the tile after next.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) slot_write(cur, grid + ticket);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S4: the intermediate is complete, the next patch has landed

    // ---- D: conv2 on my 32 output pixels (waves 0..5), bias + residual, ReLU, store.  The accumulators take this tile's
    // residual; the registers then take the next tile's, from its landed fp32 patch, before waves 6 and 7 split it (S5).
    if (has_u2) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        acc[4 * g4 + 0] = (b2r[g4].x + __uint_as_float(rr[g4].x)) * acc2_scale;
        acc[4 * g4 + 1] = (b2r[g4].y + __uint_as_float(rr[g4].y)) * acc2_scale;
        acc[4 * g4 + 2] = (b2r[g4].z + __uint_as_float(rr[g4].z)) * acc2_scale;
        acc[4 * g4 + 3] = (b2r[g4].w + __uint_as_float(rr[g4].w)) * acc2_scale;
      }
      asm volatile("" : "+v"(acc));
      if (has_next) read_residual(smem + R_OFF + (cur ^ 1) * B_R_BYTES);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S5: the next tile's residuals are in registers
    if (has_u2) {
      u32x4b prX[2], prY[2], wqX[2], wqY[2];
      B_READ(X, region, w2_bytes, p2base, BI_W, 0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        B_READ(Y, region, w2_bytes, p2base, BI_W, tap, 1); B_PIN(); B_MFMA(X); B_PIN();
        if (tap < 8) B_READ(X, region, w2_bytes, p2base, BI_W, tap + 1, 0);
        B_PIN(); B_MFMA(Y); B_PIN();
      }
      const int m = (c_row0 + c_y0 + oy) * W + c_x0 + ox;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x4b pk;
        pk.x = __float_as_uint(fmaxf(acc[4 * g4 + 0] * acc2_unscale, 0.f));
        pk.y = __float_as_uint(fmaxf(acc[4 * g4 + 1] * acc2_unscale, 0.f));
        pk.z = __float_as_uint(fmaxf(acc[4 * g4 + 2] * acc2_unscale, 0.f));
        pk.w = __float_as_uint(fmaxf(acc[4 * g4 + 3] * acc2_unscale, 0.f));
        out_bits = max(max(out_bits, max(pk.x & 0x7FFFFFFFu, pk.y & 0x7FFFFFFFu)), max(pk.z & 0x7FFFFFFFu, pk.w & 0x7FFFFFFFu));
        __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, (unsigned)(m * C + 8 * g4 + 4 * fh) * 4u, 0, 0);
      }
    } else if (has_next) {
      // ---- waves 6, 7 (no conv2 block): split the next tile's patch, landed before S4, in place
      char* nreg = smem + R_OFF + (cur ^ 1) * B_R_BYTES;
#pragma unroll 1
      for (int row = tid - 64 * U2; row < BP_PIX; row += 64 * (B_WAVES - U2)) convert_row(nreg, row);
    }
    if (!has_next) break;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();            // S2: the next patch is split; conv2's reads of this region are done
    const int next2 = slot_read(cur);        // written before S4
    tile = next;
    next = next2;
    c_row0 = n_row0; c_y0 = n_y0; c_x0 = n_x0;
    cur ^= 1;
  }
  if (p.out_max) publish_abs_max(p.out_max, out_bits);
#undef B_PIN
#undef B_READ
#undef B_MFMA
}

bool conv_block32_applicable(const BlockLaunch& b) {
  return b.in && b.out && b.w1_split && b.w2_split && b.bias1 && b.bias2 && b.in_max && b.tile_counter && b.num_cu > 0 &&
         b.unscale_w1 > 0.f && b.unscale_w2 > 0.f && b.n_img > 0 && b.H % BT_Y == 0 && b.W % BT_X == 0 &&
         (size_t)b.n_img * b.H * b.W * 32 * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_block32(const BlockLaunch& b, hipStream_t s) {
  if (!conv_block32_applicable(b)) return hipErrorInvalidValue;
  const int tiles_x = b.W / BT_X, tiles_per_img = tiles_x * (b.H / BT_Y);
  const int n_tiles = b.n_img * tiles_per_img;
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (b.device >= 0 && b.device < 64) ? 1ull << b.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_block32_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = b.num_cu;           // one 512-thread workgroup per CU (LDS: 156 KB)
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(conv_block32_kernel, dim3(grid), dim3(64 * B_WAVES), B_LDS, s, b, tiles_x, tiles_per_img, n_tiles);
  return hipGetLastError();
}

}  // namespace ut
