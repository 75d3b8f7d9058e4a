// 3x3 / stride-1 / 64 -> 64-channel convolution (five of layer2's six convolutions, lib/models/backbone_resnet.py:56-72 at
// 24x24x64) in the split-fp16 arithmetic of conv_split.hip, with the WEIGHTS RESIDENT IN REGISTERS and K SPLIT ACROSS THE TWO
// WAVES OF A SIMD.
//
// conv_c64r.hip keeps a 32-channel output block's 32 x 576 weights in the 512 registers of ONE wave per SIMD: no weight stream,
// no chunk synchronisation - but a single in-order wave exposes every latency (LDS round trips, the in-place split of the next
// patch, the epilogue: 60 % of a tile's time had no MFMA in flight).  Here a workgroup has EIGHT waves, two per SIMD, and the
// two waves of a SIMD share an output block and split its K: wave ks = 0 holds the weights of input channels 0..31 (9 taps x 2
// k-steps x 2 planes = 36 fragments = 144 registers, in the accumulator half of the file), wave ks = 1 those of channels 32..63.
//   * a tile is 128 consecutive pixels (4 blocks of 32); the wave pair (cb, ph) owns output channels 32 cb .. 32 cb + 31 of the
//     blocks 2 j + ph; each wave runs 9 taps x 2 k-steps x 2 blocks x 3 products = 108 MFMAs per tile on its slice of the patch
//     with no synchronisation inside (the ninth tap's weights, which do not fit beside the other eight in the accumulator half
//     of a 256-register wave, are read from LDS where they are used);
//   * both 32-channel slices of a tile's input rows (128 + one image row and one pixel on either side: <= 192 rows) are in LDS
//     at once, the next tile's two slices stream in meanwhile (4 x 24 KB; every wave transfers six 1-KB pieces of its own
//     slice and counts them in when they have landed);
//   * wave ks = 1 starts a tile from (residual + bias) / scale in its accumulators instead of zero, hands its partial sums to
//     its partner through LDS (8 KB per pair) and requests the next tile's residual; behind its MFMAs every wave splits its own
//     slice of the next tile's patches in place; wave ks = 0 adds the partial sums, scales, clamps and stores (no load on its
//     path) after the tile's ONE barrier, while its partner is in the next tile's MFMAs: a SIMD's matrix pipe has the other
//     wave's MFMAs while one wave is in its epilogue or in the split, and LDS / global latencies of one wave are covered by the other.
// MFMAs are inline asm (operand classes and registers are this file's choice; an asm MFMA's wait states travel inside the
// statement where the compiler could put a copy next to it), the pixels are their first operand (a lane's accumulators are one
// output channel of sixteen pixels: dword accesses of the epilogue are whole 128-byte half rows).  Same interface and tensors as
// conv_split_kernel<256, 64, 8, 1, true>; the sum over K is taken as (slice 0) + (slice 1, residual, bias) instead of one running
// sum with bias and residual added last: results agree with that kernel's to fp32 rounding, not bit for bit; deterministic.
#include <atomic>

#include "ut_kernels.h"

namespace ut {
namespace {

typedef float f32x16k __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2k __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_chark;

constexpr int K_BM = 128;                 // pixels per tile
constexpr int K_HROWS = 192;              // patch rows per slice (128 + 2 * (image width + 1) <= 192: width <= 31)
constexpr int K_STAGE = K_HROWS * 128;    // one slice patch: 24 KB; buffer (tile parity) * 2 + slice
constexpr int K_ZROW = 4 * K_STAGE;       // 256 bytes of zeros
constexpr int K_MASK = K_ZROW + 256;      // per-pixel-of-the-image 9-bit tap validity masks (u32), up to K_MAXHW pixels
constexpr int K_MAXHW = 1024;
constexpr int K_XCH = K_MASK + K_MAXHW * 4;          // partial sums of the ks = 1 waves: 4 pairs x 2 blocks x 4 KB
constexpr int K_W8 = K_XCH + 4 * 2 * 4096;           // the ninth tap's weight fragments of (cb, ks): 4 x 4 KB
constexpr int K_CNT = K_W8 + 4 * 4096;               // three counters (monotonic): patches landed [slice 0], [slice 1]; exchange area read
constexpr int K_LDS = K_CNT + 32;
constexpr int K_PW = K_HROWS / 8 / 4;     // 1-KB pieces of a slice patch per wave of that slice: 6
constexpr unsigned K_HOOB = 0x80000000u;
static_assert(K_ZROW % 256 == 0, "zero block bank-row aligned");

__device__ __forceinline__ void k_dma(u32x4k rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc), "s"(soffset)
      : "memory");
}
__device__ __forceinline__ u32x4k k_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4k r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void k_split_scaled(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2k h = __builtin_bit_cast(f16x2k, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

// Accumulator-file registers of weight fragment f = tap * 4 + k-step * 2 + plane: a[4 f : 4 f + 3], as inline-asm constraints.  Every
// use pins the fragment to the same physical registers, so the compiler knows they are occupied (registers it is not told about it
// hands to other values) and has no reason to move them (with plain "a" constraints its allocator kept shuffling and spilling them).
#define K_AR_0_0_0 "{a[0:3]}"
#define K_AR_0_0_1 "{a[4:7]}"
#define K_AR_0_1_0 "{a[8:11]}"
#define K_AR_0_1_1 "{a[12:15]}"
#define K_AR_1_0_0 "{a[16:19]}"
#define K_AR_1_0_1 "{a[20:23]}"
#define K_AR_1_1_0 "{a[24:27]}"
#define K_AR_1_1_1 "{a[28:31]}"
#define K_AR_2_0_0 "{a[32:35]}"
#define K_AR_2_0_1 "{a[36:39]}"
#define K_AR_2_1_0 "{a[40:43]}"
#define K_AR_2_1_1 "{a[44:47]}"
#define K_AR_3_0_0 "{a[48:51]}"
#define K_AR_3_0_1 "{a[52:55]}"
#define K_AR_3_1_0 "{a[56:59]}"
#define K_AR_3_1_1 "{a[60:63]}"
#define K_AR_4_0_0 "{a[64:67]}"
#define K_AR_4_0_1 "{a[68:71]}"
#define K_AR_4_1_0 "{a[72:75]}"
#define K_AR_4_1_1 "{a[76:79]}"
#define K_AR_5_0_0 "{a[80:83]}"
#define K_AR_5_0_1 "{a[84:87]}"
#define K_AR_5_1_0 "{a[88:91]}"
#define K_AR_5_1_1 "{a[92:95]}"
#define K_AR_6_0_0 "{a[96:99]}"
#define K_AR_6_0_1 "{a[100:103]}"
#define K_AR_6_1_0 "{a[104:107]}"
#define K_AR_6_1_1 "{a[108:111]}"
#define K_AR_7_0_0 "{a[112:115]}"
#define K_AR_7_0_1 "{a[116:119]}"
#define K_AR_7_1_0 "{a[120:123]}"
#define K_AR_7_1_1 "{a[124:127]}"
// (tap 8 is not register-resident: these only let the discarded branch of an `if constexpr` parse)
#define K_AR_8_0_0 "v"
#define K_AR_8_0_1 "v"
#define K_AR_8_1_0 "v"
#define K_AR_8_1_1 "v"
#define K_ARF_0 "{a[0:3]}"
#define K_ARF_1 "{a[4:7]}"
#define K_ARF_2 "{a[8:11]}"
#define K_ARF_3 "{a[12:15]}"
#define K_ARF_4 "{a[16:19]}"
#define K_ARF_5 "{a[20:23]}"
#define K_ARF_6 "{a[24:27]}"
#define K_ARF_7 "{a[28:31]}"
#define K_ARF_8 "{a[32:35]}"
#define K_ARF_9 "{a[36:39]}"
#define K_ARF_10 "{a[40:43]}"
#define K_ARF_11 "{a[44:47]}"
#define K_ARF_12 "{a[48:51]}"
#define K_ARF_13 "{a[52:55]}"
#define K_ARF_14 "{a[56:59]}"
#define K_ARF_15 "{a[60:63]}"
#define K_ARF_16 "{a[64:67]}"
#define K_ARF_17 "{a[68:71]}"
#define K_ARF_18 "{a[72:75]}"
#define K_ARF_19 "{a[76:79]}"
#define K_ARF_20 "{a[80:83]}"
#define K_ARF_21 "{a[84:87]}"
#define K_ARF_22 "{a[88:91]}"
#define K_ARF_23 "{a[92:95]}"
#define K_ARF_24 "{a[96:99]}"
#define K_ARF_25 "{a[100:103]}"
#define K_ARF_26 "{a[104:107]}"
#define K_ARF_27 "{a[108:111]}"
#define K_ARF_28 "{a[112:115]}"
#define K_ARF_29 "{a[116:119]}"
#define K_ARF_30 "{a[120:123]}"
#define K_ARF_31 "{a[124:127]}"

}  // namespace

__global__ __launch_bounds__(512, 1) void conv_c64k_kernel(ConvLaunch p, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_chark*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave & 3;                 // waves grp and grp + 4 are the two waves of SIMD grp: one output block, K split
  const int ks = wave >> 2;                 // my 32 input channels: 32 ks .. 32 ks + 31
  const int cb = grp & 1;                   // the pair's 32 output channels: 32 cb .. 32 cb + 31
  const int ph = grp >> 1;                  // the pair's pixel blocks of a tile: 2 j + ph, j = 0, 1
  const int fr = lane & 31, fh = lane >> 5;
  const int wimg = p.W;
  const int hw = p.H * p.W;
  const int M = p.n_img * hw;
  constexpr int CIN = 64, COUT = 64;

  float x_scale = 1.f, x_unscale = 1.f;
  if (p.in_max) {
    bool ok;
    split_act_scale(p.in_max, p.in_obs, x_scale, x_unscale, ok);
    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);
  }
  const float tot_unscale = p.split_unscale * x_unscale;

  const u32x4k a_words = k_rsrc(p.in, (unsigned)((size_t)M * CIN * sizeof(float)));
  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.bias), 0, p.res ? (int)((size_t)M * COUT * sizeof(float)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)M * COUT * sizeof(float)), 0x00020000);

  // ---- my weights: group cb of ConvLaunch::w_split ([cout / 32][chunk 18][k-step 2][plane 2][lane 64][8 halves]; chunk = slice * 9 +
  // tap), the nine chunks of slice ks: fragment f = tap * 4 + k-step * 2 + plane.  Taps 0..7 live in the accumulator half of the
  // register file, each pinned to its own registers a[4 f : 4 f + 3] at every use (K_AR_* above); the ninth tap's four fragments
  // are read from LDS where they are used (4 KB per tile and wave).
  u32x4k wa[32];
  {
    const char* wg = reinterpret_cast<const char*>(p.w_split) + ((size_t)cb * 18 + (size_t)ks * 9) * 4096 + lane * 16;
    // (sixteen requests and their wait in ONE statement: the compiler takes an asm's outputs as ready where the statement ends, and
    // would spill or copy a register whose load has not landed)
    asm volatile(
        "global_load_dwordx4 %0, %16, off offset:0\n\t"
        "global_load_dwordx4 %1, %16, off offset:1024\n\t"
        "global_load_dwordx4 %2, %16, off offset:2048\n\t"
        "global_load_dwordx4 %3, %16, off offset:3072\n\t"
        "global_load_dwordx4 %4, %17, off offset:0\n\t"
        "global_load_dwordx4 %5, %17, off offset:1024\n\t"
        "global_load_dwordx4 %6, %17, off offset:2048\n\t"
        "global_load_dwordx4 %7, %17, off offset:3072\n\t"
        "global_load_dwordx4 %8, %18, off offset:0\n\t"
        "global_load_dwordx4 %9, %18, off offset:1024\n\t"
        "global_load_dwordx4 %10, %18, off offset:2048\n\t"
        "global_load_dwordx4 %11, %18, off offset:3072\n\t"
        "global_load_dwordx4 %12, %19, off offset:0\n\t"
        "global_load_dwordx4 %13, %19, off offset:1024\n\t"
        "global_load_dwordx4 %14, %19, off offset:2048\n\t"
        "global_load_dwordx4 %15, %19, off offset:3072\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&" K_ARF_0(wa[0]), "=&" K_ARF_1(wa[1]), "=&" K_ARF_2(wa[2]), "=&" K_ARF_3(wa[3]), "=&" K_ARF_4(wa[4]), "=&" K_ARF_5(wa[5]), "=&" K_ARF_6(wa[6]), "=&" K_ARF_7(wa[7]), "=&" K_ARF_8(wa[8]), "=&" K_ARF_9(wa[9]), "=&" K_ARF_10(wa[10]), "=&" K_ARF_11(wa[11]), "=&" K_ARF_12(wa[12]), "=&" K_ARF_13(wa[13]), "=&" K_ARF_14(wa[14]), "=&" K_ARF_15(wa[15])
        : "v"(wg + 0 * 4096), "v"(wg + 1 * 4096), "v"(wg + 2 * 4096), "v"(wg + 3 * 4096)
        : "memory");
    asm volatile(
        "global_load_dwordx4 %0, %16, off offset:0\n\t"
        "global_load_dwordx4 %1, %16, off offset:1024\n\t"
        "global_load_dwordx4 %2, %16, off offset:2048\n\t"
        "global_load_dwordx4 %3, %16, off offset:3072\n\t"
        "global_load_dwordx4 %4, %17, off offset:0\n\t"
        "global_load_dwordx4 %5, %17, off offset:1024\n\t"
        "global_load_dwordx4 %6, %17, off offset:2048\n\t"
        "global_load_dwordx4 %7, %17, off offset:3072\n\t"
        "global_load_dwordx4 %8, %18, off offset:0\n\t"
        "global_load_dwordx4 %9, %18, off offset:1024\n\t"
        "global_load_dwordx4 %10, %18, off offset:2048\n\t"
        "global_load_dwordx4 %11, %18, off offset:3072\n\t"
        "global_load_dwordx4 %12, %19, off offset:0\n\t"
        "global_load_dwordx4 %13, %19, off offset:1024\n\t"
        "global_load_dwordx4 %14, %19, off offset:2048\n\t"
        "global_load_dwordx4 %15, %19, off offset:3072\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&" K_ARF_16(wa[16]), "=&" K_ARF_17(wa[17]), "=&" K_ARF_18(wa[18]), "=&" K_ARF_19(wa[19]), "=&" K_ARF_20(wa[20]), "=&" K_ARF_21(wa[21]), "=&" K_ARF_22(wa[22]), "=&" K_ARF_23(wa[23]), "=&" K_ARF_24(wa[24]), "=&" K_ARF_25(wa[25]), "=&" K_ARF_26(wa[26]), "=&" K_ARF_27(wa[27]), "=&" K_ARF_28(wa[28]), "=&" K_ARF_29(wa[29]), "=&" K_ARF_30(wa[30]), "=&" K_ARF_31(wa[31])
        : "v"(wg + 4 * 4096), "v"(wg + 5 * 4096), "v"(wg + 6 * 4096), "v"(wg + 7 * 4096)
        : "memory");
    if (ph == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<u32x4k*>(smem + K_W8 + (cb * 2 + ks) * 4096 + q * 1024 + lane * 16) =
            *reinterpret_cast<const u32x4k*>(wg + 8 * 4096 + q * 1024);
    }
  }
  const unsigned w8 = (unsigned)(K_W8 + (cb * 2 + ks) * 4096 + lane * 16);

  // ---- zero block, tap-validity masks of every pixel position of an image
  if (tid < 16) *reinterpret_cast<u32x4k*>(smem + K_ZROW + tid * 16) = u32x4k{0, 0, 0, 0};
  if (tid == 16) *reinterpret_cast<u32x4k*>(smem + K_CNT) = u32x4k{0, 0, 0, 0};
  for (int pos = tid; pos < hw; pos += 512) {
    const int y = pos / wimg, x = pos - y * wimg;
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const bool ok = (unsigned)(y + t / 3 - 1) < (unsigned)p.H && (unsigned)(x + t % 3 - 1) < (unsigned)wimg;
      mk |= (ok ? 1u : 0u) << t;
    }
    *reinterpret_cast<unsigned*>(smem + K_MASK + pos * 4) = mk;
  }

  // ---- the patch stream: my k-th tile's slices live in buffers (k & 1) * 2 + slice
  const int grid = gridDim.x;
  const int my_tiles = (n_tiles - (int)blockIdx.x + grid - 1) / grid;       // tiles blockIdx.x, + grid, ...
  // (lane_o / grp_o: per-tile opaque copies, so that the pieces' row / swizzle / address constants are recomputed at issue time - a
  // handful of instructions between MFMAs - instead of hoisted into registers that live across the tile)
  int lane_o = lane, grp_o = grp;
  auto issue_patch_piece = [&](int i, int tile, int buf) {       // piece grp + 4 i of MY slice of the patch of `tile`
    const int q = grp_o + 4 * i;
    const int row = 8 * q + (lane_o >> 3);
    const int pix = tile * K_BM - wimg - 1 + row;
    const bool ok = pix >= 0 && pix < M;
    const unsigned off = ok ? (unsigned)(pix * CIN + 4 * ((lane_o & 7) ^ ((row >> 1) & 7))) * 4u : K_HOOB;
    k_dma(a_words, (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_addr + (unsigned)(buf * K_STAGE + q * 1024))), off,
          (unsigned)ks * 128u);
  };
  // split landed fp32 slice patches of a tile in place (group q = k / 8 at q ^ swizzle); unit u = slice * 192 + row, u < 384
  auto convert_unit = [&](int parity, int u) {
    const int sl = u >= K_HROWS ? 1 : 0, row = u - sl * K_HROWS;
    const int sw = (row >> 1) & 7;
    char* rp = smem + (parity * 2 + sl) * K_STAGE + row * 128;
    float4 f[8];
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4) f[g4] = *reinterpret_cast<const float4*>(rp + ((g4 ^ sw) << 4));
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      unsigned a0, a1, a2, a3, b0, b1, b2, b3;
      k_split_scaled(f[2 * kg].x, f[2 * kg].y, x_scale, a0, b0);
      k_split_scaled(f[2 * kg].z, f[2 * kg].w, x_scale, a1, b1);
      k_split_scaled(f[2 * kg + 1].x, f[2 * kg + 1].y, x_scale, a2, b2);
      k_split_scaled(f[2 * kg + 1].z, f[2 * kg + 1].w, x_scale, a3, b3);
      u32x4k a, b;
      a.x = a0; a.y = a1; a.z = a2; a.w = a3;
      b.x = b0; b.y = b1; b.z = b2; b.w = b3;
      *reinterpret_cast<u32x4k*>(rp + ((kg ^ sw) << 4)) = a;
      *reinterpret_cast<u32x4k*>(rp + (((4 + kg) ^ sw) << 4)) = b;
    }
  };
  auto convert_patches = [&](int parity) {  // every thread of the workgroup (prologue)
    int t_ = tid;
    asm volatile("" : "+v"(t_));      // (opaque: the row's addresses are computed here, not kept in registers across the MFMA loop)
    if (t_ < 2 * K_HROWS) convert_unit(parity, t_);
  };
  auto convert_own_slice = [&](int parity) {  // the four waves of a slice (256 threads) split its 192 rows
    int t_ = tid & 255;
    asm volatile("" : "+v"(t_));
    if (t_ < K_HROWS) convert_unit(parity, ks * K_HROWS + t_);
  };

  // prologue: my first tile's patches, split
#pragma unroll
  for (int i = 0; i < K_PW; ++i) issue_patch_piece(i, blockIdx.x, ks);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  convert_patches(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();

  unsigned out_bits = 0;
  const float bias_v = p.bias[32 * cb + fr];      // my output channel's bias
  const float inv_unscale = 1.f / tot_unscale;    // (a power of two: exact)
  const float bias_scaled = bias_v * inv_unscale;
  // Bias and residual enter through the ks = 1 wave's ACCUMULATORS: it starts a tile from (residual + bias) x inv_unscale instead
  // of zero (requested after barrier B1 of the tile before, landed under the split of the patches), so the epilogue wave has no
  // global load on its path - it adds the two partial sums, scales, clamps and stores.
  // A block's sixteen accesses (a lane: output channel 32 cb + fr, pixels 8 (r / 4) + 4 fh + r % 4 of the block) are two per-lane
  // offsets plus constants in the instruction's offset field, which the descriptor's range check covers (a scalar offset is not
  // checked): pixels beyond the tensor (last tile) load zeros and their stores are dropped, with no compare per access.
#define K_OFF(TILE, J) ((unsigned)(((TILE) * K_BM + 32 * (2 * (J) + ph) + 4 * fh) * COUT + 32 * cb + fr) * 4u)
#define K_ROFF(OFF, R) ((((R) >> 3) ? (OFF) + 16u * COUT * 4u : (OFF)) + (unsigned)(((((R) >> 2) & 1) * 8 + ((R) & 3)) * COUT * 4))
  f32x16k acc[2];      // (ks = 1: between its hand-over and the next tile's start these registers hold the next tile's residual)
#define K_RINI(TILE)                                                                                 \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
    const unsigned off_ = K_OFF(TILE, j);                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                   \
      acc[j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, K_ROFF(off_, r), 0, 0)); \
  }
  if (ks == 1) {
    K_RINI((int)blockIdx.x)
  }
  int lrow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) lrow[j] = 32 * (2 * j + ph) + fr + wimg + 1;      // patch row of my pixel of block j, centre tap
  const unsigned xch = (unsigned)(K_XCH + grp * 8192 + lane * 16);
  const unsigned cnt_addr = smem_addr + (unsigned)K_CNT;      // + 0 / + 4: patches of slice 0 / 1 landed; + 8: exchange area read               // the pair's exchange area, my 16 bytes of a KB

#define K_PIN() __builtin_amdgcn_sched_barrier(0)
  // LDS counters between the waves of the workgroup (one lane adds; a waiter polls until the count is reached)
#define K_SIGNAL(ADDR)                                                                               \
  {                                                                                                  \
    unsigned long long keep_;                                                                        \
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0" \
                 : "=&s"(keep_) : "v"(ADDR), "v"(1u) : "memory");                                    \
  }
#define K_AWAIT(ADDR, TARGET)                                                                        \
  for (;;) {                                                                                         \
    unsigned seen_;                                                                                  \
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen_) : "v"(ADDR) : "memory"); \
    if ((int)(__builtin_amdgcn_readfirstlane(seen_) - (unsigned)(TARGET)) >= 0) break;               \
    __builtin_amdgcn_s_sleep(1);                                                                     \
  }
#define K_MFMA_A(ACC, TAP, S, PL, PXV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(PXV), K_AR_##TAP##_##S##_##PL(wa[K_W(TAP, S, PL)]))
#define K_MFMA_V(ACC, WV, PXV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(PXV), "v"(WV))
  // (the compiler knows nothing of an asm MFMA's latency and may put a register copy of the accumulator right behind - or in front of -
  // it: the wait states travel INSIDE the statement of a block's first and last MFMA of a tile)
#define K_MFMA_A_FIRST(ACC, TAP, S, PL, PXV) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(PXV), K_AR_##TAP##_##S##_##PL(wa[K_W(TAP, S, PL)]))
#define K_MFMA_V_LAST(ACC, WV, PXV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(ACC) : "v"(PXV), "v"(WV))
  // Address of the lane's 16 bytes of block J, tap TAP, k-step 0, first pieces, in the patch at byte offset BUF - or in the block of
  // zeros, on the same banks, when the tap leaves the image.  k-step 1 is that address ^ 32, the remainder pieces ^ 64.
#define K_ADDR(DST, BUF, J, TAP)                                                                     \
  {                                                                                                  \
    const int row_ = lrow_t[J] + ((TAP) / 3 - 1) * wimg + ((TAP) % 3 - 1);                           \
    const unsigned a_ = (unsigned)(BUF) + (unsigned)(row_ * 128) + (unsigned)(((fh ^ ((row_ >> 1) & 7))) << 4); \
    DST = ((rmask[J] >> (TAP)) & 1u) ? a_ : (unsigned)K_ZROW + (a_ & 255u);                          \
  }
#define K_LOAD(DST, ADDR, S)                                                                         \
  {                                                                                                  \
    DST[0] = *reinterpret_cast<const u32x4k*>(smem + ((ADDR) ^ (32u * (S))));                        \
    DST[1] = *reinterpret_cast<const u32x4k*>(smem + ((ADDR) ^ (32u * (S)) ^ 64u));                  \
  }
#define K_LEAD() asm volatile("s_nop 3")
#define K_DRAIN() asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]))
#define K_W(TAP, S, PL) (((TAP) * 4 + (S) * 2 + (PL)) & 31)
  // One k-step: for each of my two pixel blocks three MFMAs (weights' first pieces x pixels' remainders, weights' remainders x
  // pixels' first pieces, first x first - small terms first, like conv_split.hip).  The other instructions sit between them: the
  // four fragment reads of the NEXT k-step right behind the step's first two MFMAs (five and seven MFMAs of this wave - twice that
  // in time, with the SIMD's other wave interleaved - before their first use: a wave has only six MFMAs per k-step to cover an LDS
  // round trip with), the addresses of the next tap's fragments one k-step before those reads (S = 0; address sets alternate by tap
  // parity), a transfer piece of the next tile's patch behind the last one.
  // MFMA W (0, 1, 2) of block J in k-step (TAP, S)
#define K_MM(TAP, S, CUR, J, W)                                                                      \
  {                                                                                                  \
    if constexpr ((TAP) < 8) {                                                                       \
      if constexpr ((W) == 0) {                                                                      \
        if constexpr ((TAP) == 0 && (S) == 0) { K_MFMA_A_FIRST(acc[J], TAP, S, 0, CUR[J][1]); }      \
        else { K_MFMA_A(acc[J], TAP, S, 0, CUR[J][1]); }                                             \
      } else if constexpr ((W) == 1) { K_MFMA_A(acc[J], TAP, S, 1, CUR[J][0]); }                     \
      else { K_MFMA_A(acc[J], TAP, S, 0, CUR[J][0]); }                                               \
    } else {                                                                                         \
      if constexpr ((W) == 0) { K_MFMA_V(acc[J], w8a_, CUR[J][1]); }                                 \
      else if constexpr ((W) == 1) { K_MFMA_V(acc[J], w8b_, CUR[J][0]); }                            \
      else if constexpr ((S) == 1) { K_MFMA_V_LAST(acc[J], w8a_, CUR[J][0]); }                       \
      else { K_MFMA_V(acc[J], w8a_, CUR[J][0]); }                                                    \
    }                                                                                                \
    K_PIN();                                                                                         \
  }
  // k-step (TAP, S) consumes CUR and reads the fragments of the k-step AFTER NEXT - (TAP + 1, S) - into FAR (three register sets in
  // rotation); ADN holds the addresses of tap TAP + 1 (computed a k-step earlier), ADF receives those of tap TAP + 2 (S = 1)
#define K_STEP(TAP, S, CUR, FAR, PIECE, ADN, ADF)                                                    \
  {                                                                                                  \
    u32x4k w8a_, w8b_;                                                                               \
    if constexpr ((TAP) == 8) {                                                                      \
      w8a_ = *reinterpret_cast<const u32x4k*>(smem + w8 + ((S) * 2 + 0) * 1024);                     \
      w8b_ = *reinterpret_cast<const u32x4k*>(smem + w8 + ((S) * 2 + 1) * 1024);                     \
    }                                                                                                \
    K_MM(TAP, S, CUR, 0, 0)                                                                          \
    if constexpr ((TAP) < 8) K_LOAD(FAR[0], ADN[0], S);                                              \
    K_PIN();                                                                                         \
    K_MM(TAP, S, CUR, 0, 1)                                                                          \
    if constexpr ((TAP) < 8) K_LOAD(FAR[1], ADN[1], S);                                              \
    K_PIN();                                                                                         \
    K_MM(TAP, S, CUR, 0, 2)                                                                          \
    K_MM(TAP, S, CUR, 1, 0)                                                                          \
    if constexpr ((S) == 1 && (TAP) < 7) K_ADDR(ADF[0], rbuf, 0, (TAP) + 2);                         \
    K_PIN();                                                                                         \
    K_MM(TAP, S, CUR, 1, 1)                                                                          \
    if constexpr ((S) == 1 && (TAP) < 7) K_ADDR(ADF[1], rbuf, 1, (TAP) + 2);                         \
    K_PIN();                                                                                         \
    K_MM(TAP, S, CUR, 1, 2)                                                                          \
    if ((PIECE) < K_PW && dma_on) issue_patch_piece(PIECE, f_tile, f_buf);                           \
    if ((PIECE) == 77 && dma_on) {      /* my six pieces (issued 8+ k-steps ago) have landed: tell the splitting waves */ \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
      K_SIGNAL(cnt_addr + 4u * (unsigned)ks);                                                        \
    }                                                                                                \
    K_PIN();                                                                                         \
  }

  for (int k = 0; k < my_tiles; ++k) {
    const int tile = blockIdx.x + k * grid;
    const unsigned rbuf = (unsigned)(((k & 1) * 2 + ks) * K_STAGE);
    // the next tile's patches stream in under this tile's MFMAs, into the buffers the tile before this one read (every wave is past
    // that tile's last read: barrier B1 of the previous iteration)
    const bool dma_on = k + 1 < my_tiles;
    const int f_tile = tile + grid, f_buf = ((k + 1) & 1) * 2 + ks;
    asm volatile("" : "+v"(lane_o));
    asm volatile("" : "+s"(grp_o));
    // my pixels' tap masks for this tile (position in the image = pixel index modulo the image size) and opaque per-tile row
    // bases (keeps the per-tap address arithmetic inside the loop instead of hoisted into registers that live across it)
    unsigned rmask[2];
    int lrow_t[2];
    {
      const int pos0 = (int)(((long)tile * K_BM) % hw);      // wave-uniform
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = tile * K_BM + 32 * (2 * j + ph) + fr;
        int pos = pos0 + 32 * (2 * j + ph) + fr;
        pos = pos >= hw ? pos - hw : pos;
        const unsigned mk = *reinterpret_cast<const unsigned*>(smem + K_MASK + pos * 4);
        rmask[j] = m < M ? mk : 0u;
        lrow_t[j] = lrow[j];
        asm volatile("" : "+v"(lrow_t[j]));
      }
    }
    if (ks == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)      // (asm: the compiler vectorises this loop with a 32-register splat of bias_scaled that lives across the kernel)
          asm("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[j][e]) : "v"(inv_unscale), "v"(bias_scaled));
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[j]));
    {
      u32x4k pxA[2][2], pxB[2][2], pxC[2][2];
      unsigned adE[2], adO[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        K_ADDR(adE[j], rbuf, j, 0);
        K_LOAD(pxA[j], adE[j], 0);
        K_LOAD(pxB[j], adE[j], 1);
        K_ADDR(adO[j], rbuf, j, 1);
      }
      K_LEAD();
      K_STEP(0, 0, pxA, pxC, 0, adO, adE)  K_STEP(0, 1, pxB, pxA, 1, adO, adE)
      K_STEP(1, 0, pxC, pxB, 2, adE, adO)  K_STEP(1, 1, pxA, pxC, 3, adE, adO)
      K_STEP(2, 0, pxB, pxA, 4, adO, adE)  K_STEP(2, 1, pxC, pxB, 5, adO, adE)
      K_STEP(3, 0, pxA, pxC, 99, adE, adO)  K_STEP(3, 1, pxB, pxA, 99, adE, adO)
      K_STEP(4, 0, pxC, pxB, 99, adO, adE)  K_STEP(4, 1, pxA, pxC, 99, adO, adE)
      K_STEP(5, 0, pxB, pxA, 99, adE, adO)  K_STEP(5, 1, pxC, pxB, 99, adE, adO)
      K_STEP(6, 0, pxA, pxC, 99, adO, adE)  K_STEP(6, 1, pxB, pxA, 99, adO, adE)
      K_STEP(7, 0, pxC, pxB, 77, adE, adO)  K_STEP(7, 1, pxA, pxC, 99, adE, adO)
      K_STEP(8, 0, pxB, pxA, 99, adO, adE)  K_STEP(8, 1, pxC, pxB, 99, adO, adE)
    }
    K_DRAIN();

    // Behind its MFMAs a wave splits ITS slice of the next tile's patches (landed: the slice's counter - its four waves transferred
    // it); the ks = 1 waves first hand over their partial sums and afterwards request the next tile's residual into the freed
    // accumulators.  ONE barrier per tile: every wave's reads of this tile's patches are done, the next tile's patches are split,
    // the partial sums are in place.
    if (ks == 1) {
      K_AWAIT(cnt_addr + 8u, 4 * k)       // every epilogue wave has read the exchange area of the tile before
#pragma unroll
      for (int j = 0; j < 2; ++j)        // block j at 4 KB j, accumulator registers 4 q .. 4 q + 3 at 1 KB q
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v;
          v.x = acc[j][4 * q + 0]; v.y = acc[j][4 * q + 1]; v.z = acc[j][4 * q + 2]; v.w = acc[j][4 * q + 3];
          *reinterpret_cast<float4*>(smem + xch + j * 4096 + q * 1024) = v;
        }
    }
    if (dma_on) {
      K_AWAIT(cnt_addr + 4u * (unsigned)ks, 4 * (k + 1))
      convert_own_slice((k + 1) & 1);
      if (ks == 1) {
        K_RINI(f_tile)      // (behind the split: the accumulators are free registers for it; the requests land while I wait at the barrier)
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (ks == 0) {
      // ---- epilogue: 1 / (weight scale x activation scale) x (my sum + my partner's, which started from residual + bias), ReLU,
      // store: one dword access per accumulator register covers two whole 128-byte half rows of [pixel][32 cb .. 32 cb + 31].
      const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();
      const bool ragged = tile * K_BM + K_BM > M;      // wave-uniform: only then the maximum needs the per-pixel mask
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 pp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pp[q] = *reinterpret_cast<const float4*>(smem + xch + j * 4096 + q * 1024);
        float o[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[4 * q + 0] = acc[j][4 * q + 0] + pp[q].x; o[4 * q + 1] = acc[j][4 * q + 1] + pp[q].y;
          o[4 * q + 2] = acc[j][4 * q + 2] + pp[q].z; o[4 * q + 3] = acc[j][4 * q + 3] + pp[q].w;
        }
        unsigned mx = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o[r] = fmaxf(o[r] * tot_unscale, floor_v);
          mx = max(mx, __float_as_uint(o[r]) & 0x7FFFFFFFu);
        }
        if (ragged) {
          mx = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            mx = max(mx, tile * K_BM + 32 * (2 * j + ph) + 8 * (r >> 2) + 4 * fh + (r & 3) < M ? __float_as_uint(o[r]) & 0x7FFFFFFFu : 0u);
        }
        out_bits = max(out_bits, mx);
        const unsigned off_j = K_OFF(tile, j);
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[r]), o_rsrc, K_ROFF(off_j, r), 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my reads of the exchange area are done: my partner may write the next tile's
      K_SIGNAL(cnt_addr + 8u);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.out_max) publish_abs_max(p.out_max, out_bits);
#undef K_PIN
#undef K_SIGNAL
#undef K_AWAIT
#undef K_MFMA_A
#undef K_MFMA_V
#undef K_MFMA_A_FIRST
#undef K_MFMA_V_LAST
#undef K_ADDR
#undef K_LOAD
#undef K_LEAD
#undef K_DRAIN
#undef K_W
#undef K_STEP
#undef K_MM
#undef K_OFF
#undef K_ROFF
#undef K_RINI
}

bool conv_c64k_applicable(const ConvLaunch& c) {
  return !(c.no_resident & 1) && c.w_split && c.split_unscale > 0.f && c.ksize == 3 && c.stride == 1 && c.pad == 1 && c.cin == 64 && c.cout_store == 64 &&
         c.cslice == 32 && c.k_pad == 576 && c.cout_pad >= 64 && !c.out_nchw && c.splits == 0 && c.W <= 31 && c.H == c.Ho &&
         c.W == c.Wo && c.H * c.W <= K_MAXHW && c.H * c.W >= K_BM && c.num_cu > 0 &&
         (size_t)c.n_img * c.H * c.W * 64 * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_c64k(const ConvLaunch& c, hipStream_t s) {
  if (!conv_c64k_applicable(c)) return hipErrorInvalidValue;
  const long M = (long)c.n_img * c.H * c.W;
  const int n_tiles = (int)((M + K_BM - 1) / K_BM);
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c64k_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, K_LDS);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;           // one 512-thread workgroup per CU, two waves per SIMD; tiles are dealt round robin (they all cost the same)
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(conv_c64k_kernel, dim3(grid), dim3(512), K_LDS, s, c, n_tiles);
  return hipGetLastError();
}

}  // namespace ut
