// The stride-2 entry of layer2 (lib/models/backbone_resnet.py:56-72 with a downsample branch, at 48x48x32 -> 24x24x64) in the
// split-fp16 arithmetic of conv_split.hip: the block's FIRST convolution (3x3, stride 2, 32 -> 64 channels, BatchNorm, ReLU) and
// its SHORTCUT (1x1, stride 2, 32 -> 64, BatchNorm) from ONE pass over the input, weights resident in registers.
//
// As two launches the 3x3 went through the gather instantiation of conv_split_kernel (every tap re-fetches and re-splits its
// pixels: 0.55 ms, the slowest split launch per FLOP) and the shortcut through the fp32 kernel (0.23 ms, bound by its bytes).
// Here a tile is 4 output rows of an image (96 pixels = 3 blocks of 32); its 9 input rows (432 pixels x 32 channels, one 32-channel
// slice: the whole K) come into LDS once by LDS-DMA, double buffered across tiles, and are split in place; six of the workgroup's
// eight waves each own (32 output channels) x (one pixel block): 9 taps x 2 k-steps x 3 products = 54 MFMAs on register-resident
// weights (taps 0..7 pinned in the accumulator half, the ninth tap from LDS, as in conv_c64k.hip) plus 6 MFMAs of the shortcut on
// the centre tap's fragments, which are in registers at that moment anyway (its 2 x 2 weight fragments: 16 registers).  The other
// two waves transfer and split (with a share taken by the six behind their MFMAs).  One barrier per tile; a counter in LDS says
// when every wave's pieces of the next patch have landed.  Both outputs leave as whole 128-byte half rows (a lane owns one output
// channel of sixteen pixels).  Same tensors and arithmetic as the two launches; the 3x3's output word (max |out|) is published.
#include <atomic>

#include "ut_kernels.h"

namespace ut {
namespace {

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2s __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_chars;


constexpr int S_OPIX = 96;                // output pixels per tile: 4 rows of 24
constexpr int S_PROWS = 432;              // patch rows (input pixels) per tile: 9 rows of <= 48
constexpr int S_STAGE = S_PROWS * 128;    // 54 KB; buffer = tile parity
constexpr int S_ZROW = 2 * S_STAGE;       // 256 bytes of zeros
constexpr int S_W8 = S_ZROW + 256;        // the ninth tap's weight fragments of cb: 2 x 4 KB
constexpr int S_CNT = S_W8 + 2 * 4096;    // counter: pieces of the next patch landed (monotonic, 8 per tile)
constexpr int S_LDS = S_CNT + 32;
constexpr unsigned S_HOOB = 0x80000000u;
static_assert(S_ZROW % 256 == 0, "zero block bank-row aligned");

__device__ __forceinline__ void s2_dma(u32x4s rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc), "s"(soffset)
      : "memory");
}
__device__ __forceinline__ u32x4s s2_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4s r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void s2_split_scaled(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2s h = __builtin_bit_cast(f16x2s, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

// Accumulator-file registers of weight fragment f = tap * 4 + k-step * 2 + plane: a[4 f : 4 f + 3], as inline-asm constraints.  Every
// use pins the fragment to the same physical registers, so the compiler knows they are occupied (registers it is not told about it
// hands to other values) and has no reason to move them (with plain "a" constraints its allocator kept shuffling and spilling them).
#define S_AR_0_0_0 "{a[0:3]}"
#define S_AR_0_0_1 "{a[4:7]}"
#define S_AR_0_1_0 "{a[8:11]}"
#define S_AR_0_1_1 "{a[12:15]}"
#define S_AR_1_0_0 "{a[16:19]}"
#define S_AR_1_0_1 "{a[20:23]}"
#define S_AR_1_1_0 "{a[24:27]}"
#define S_AR_1_1_1 "{a[28:31]}"
#define S_AR_2_0_0 "{a[32:35]}"
#define S_AR_2_0_1 "{a[36:39]}"
#define S_AR_2_1_0 "{a[40:43]}"
#define S_AR_2_1_1 "{a[44:47]}"
#define S_AR_3_0_0 "{a[48:51]}"
#define S_AR_3_0_1 "{a[52:55]}"
#define S_AR_3_1_0 "{a[56:59]}"
#define S_AR_3_1_1 "{a[60:63]}"
#define S_AR_4_0_0 "{a[64:67]}"
#define S_AR_4_0_1 "{a[68:71]}"
#define S_AR_4_1_0 "{a[72:75]}"
#define S_AR_4_1_1 "{a[76:79]}"
#define S_AR_5_0_0 "{a[80:83]}"
#define S_AR_5_0_1 "{a[84:87]}"
#define S_AR_5_1_0 "{a[88:91]}"
#define S_AR_5_1_1 "{a[92:95]}"
#define S_AR_6_0_0 "{a[96:99]}"
#define S_AR_6_0_1 "{a[100:103]}"
#define S_AR_6_1_0 "{a[104:107]}"
#define S_AR_6_1_1 "{a[108:111]}"
#define S_AR_7_0_0 "{a[112:115]}"
#define S_AR_7_0_1 "{a[116:119]}"
#define S_AR_7_1_0 "{a[120:123]}"
#define S_AR_7_1_1 "{a[124:127]}"
// (tap 8 is not register-resident: these only let the discarded branch of an `if constexpr` parse)
#define S_AR_8_0_0 "v"
#define S_AR_8_0_1 "v"
#define S_AR_8_1_0 "v"
#define S_AR_8_1_1 "v"
#define S_ARF_0 "{a[0:3]}"
#define S_ARF_1 "{a[4:7]}"
#define S_ARF_2 "{a[8:11]}"
#define S_ARF_3 "{a[12:15]}"
#define S_ARF_4 "{a[16:19]}"
#define S_ARF_5 "{a[20:23]}"
#define S_ARF_6 "{a[24:27]}"
#define S_ARF_7 "{a[28:31]}"
#define S_ARF_8 "{a[32:35]}"
#define S_ARF_9 "{a[36:39]}"
#define S_ARF_10 "{a[40:43]}"
#define S_ARF_11 "{a[44:47]}"
#define S_ARF_12 "{a[48:51]}"
#define S_ARF_13 "{a[52:55]}"
#define S_ARF_14 "{a[56:59]}"
#define S_ARF_15 "{a[60:63]}"
#define S_ARF_16 "{a[64:67]}"
#define S_ARF_17 "{a[68:71]}"
#define S_ARF_18 "{a[72:75]}"
#define S_ARF_19 "{a[76:79]}"
#define S_ARF_20 "{a[80:83]}"
#define S_ARF_21 "{a[84:87]}"
#define S_ARF_22 "{a[88:91]}"
#define S_ARF_23 "{a[92:95]}"
#define S_ARF_24 "{a[96:99]}"
#define S_ARF_25 "{a[100:103]}"
#define S_ARF_26 "{a[104:107]}"
#define S_ARF_27 "{a[108:111]}"
#define S_ARF_28 "{a[112:115]}"
#define S_ARF_29 "{a[116:119]}"
#define S_ARF_30 "{a[120:123]}"
#define S_ARF_31 "{a[124:127]}"

}  // namespace

__global__ __launch_bounds__(512, 1) void conv_c32s2_kernel(Stride2Launch p, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_chars*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mfma_wave = wave < 6;          // waves 6, 7: transfers and the split only
  const int cb = wave & 1;                  // my 32 output channels: 32 cb .. 32 cb + 31
  const int blk = mfma_wave ? wave >> 1 : 0;      // my pixel block of a tile
  const int fr = lane & 31, fh = lane >> 5;
  const int W = p.W, Wo = p.W / 2;
  const int tiles_per_img = p.H / 8;        // 4 output rows = 8 input rows per tile
  const int M_in = p.n_img * p.H * p.W;
  const int M_out = p.n_img * (p.H / 2) * Wo;
  constexpr int CIN = 32, COUT = 64;

  float x_scale = 1.f, x_unscale = 1.f;
  if (p.in_max) {
    bool ok;
    split_act_scale(p.in_max, p.in_obs, x_scale, x_unscale, ok);
    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);
  }
  const float unscale1 = p.unscale1 * x_unscale, unscale_d = p.unscale_d * x_unscale;

  const u32x4s a_words = s2_rsrc(p.in, (unsigned)((size_t)M_in * CIN * sizeof(float)));
  const __amdgpu_buffer_rsrc_t o1_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out1, 0, (int)((size_t)M_out * COUT * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t o2_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, (int)((size_t)M_out * COUT * sizeof(float)), 0x00020000);

  // ---- my weights: group cb of the 3x3's planes ([cout / 32][tap 9][k-step 2][plane 2][lane 64][8 halves]): fragment f = tap * 4 +
  // k-step * 2 + plane; taps 0..7 pinned to a[4 f : 4 f + 3], the ninth tap's four fragments in LDS; the shortcut's four fragments
  // (group cb of its planes, one chunk) in the vector half
  u32x4s wa[32], wd[4];
  {
    const char* wg = reinterpret_cast<const char*>(p.w1_split) + (size_t)cb * 9 * 4096 + lane * 16;
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
        : "=&" S_ARF_0(wa[0]), "=&" S_ARF_1(wa[1]), "=&" S_ARF_2(wa[2]), "=&" S_ARF_3(wa[3]), "=&" S_ARF_4(wa[4]), "=&" S_ARF_5(wa[5]), "=&" S_ARF_6(wa[6]), "=&" S_ARF_7(wa[7]), "=&" S_ARF_8(wa[8]), "=&" S_ARF_9(wa[9]), "=&" S_ARF_10(wa[10]), "=&" S_ARF_11(wa[11]), "=&" S_ARF_12(wa[12]), "=&" S_ARF_13(wa[13]), "=&" S_ARF_14(wa[14]), "=&" S_ARF_15(wa[15])
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
        : "=&" S_ARF_16(wa[16]), "=&" S_ARF_17(wa[17]), "=&" S_ARF_18(wa[18]), "=&" S_ARF_19(wa[19]), "=&" S_ARF_20(wa[20]), "=&" S_ARF_21(wa[21]), "=&" S_ARF_22(wa[22]), "=&" S_ARF_23(wa[23]), "=&" S_ARF_24(wa[24]), "=&" S_ARF_25(wa[25]), "=&" S_ARF_26(wa[26]), "=&" S_ARF_27(wa[27]), "=&" S_ARF_28(wa[28]), "=&" S_ARF_29(wa[29]), "=&" S_ARF_30(wa[30]), "=&" S_ARF_31(wa[31])
        : "v"(wg + 4 * 4096), "v"(wg + 5 * 4096), "v"(wg + 6 * 4096), "v"(wg + 7 * 4096)
        : "memory");
    if (wave < 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<u32x4s*>(smem + S_W8 + cb * 4096 + q * 1024 + lane * 16) = *reinterpret_cast<const u32x4s*>(wg + 8 * 4096 + q * 1024);
    }
    const char* dg = reinterpret_cast<const char*>(p.wd_split) + (size_t)cb * 4096 + lane * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) wd[q] = *reinterpret_cast<const u32x4s*>(dg + q * 1024);
  }
  const unsigned w8 = (unsigned)(S_W8 + cb * 4096 + lane * 16);

  if (tid < 16) *reinterpret_cast<u32x4s*>(smem + S_ZROW + tid * 16) = u32x4s{0, 0, 0, 0};
  if (tid == 16) *reinterpret_cast<u32x4s*>(smem + S_CNT) = u32x4s{0, 0, 0, 0};

  // ---- the patch stream: my k-th tile's patch lives in buffer k & 1.  Tile t = (image, 4 output rows 4 q .. 4 q + 3): its patch is the
  // 9 W consecutive input pixels from row 8 q - 1 of the image (row -1 of the first tile of an image is the last row of the image
  // before - or before the tensor: zeros by the descriptor's range check - and is never used: the taps that would read it are masked)
  const int grid = gridDim.x;
  const int my_tiles = (n_tiles - (int)blockIdx.x + grid - 1) / grid;
  const int n_pieces = (9 * W + 7) / 8;
  int lane_o = lane, wave_o = wave;
  auto patch_first_pixel = [&](int tile) {
    const int img = tile / tiles_per_img, q = tile - img * tiles_per_img;
    return img * p.H * W + (8 * q - 1) * W;
  };
  auto issue_patch_piece = [&](int i, int first_pixel, int buf) {       // piece wave + 8 i
    const int q = wave_o + 8 * i;
    if (q < n_pieces) {
      const int pos = 8 * q + (lane_o >> 3);                                   // LDS row position of my 16 bytes
      const int row = (pos & ~3) | ((pos & 1) << 1) | ((pos >> 1) & 1);        // ... holds patch row `row` (S_POS below)
      const int pix = first_pixel + row;
      const bool ok = pix >= 0 && pix < M_in && row < 9 * W;
      const unsigned off = ok ? (unsigned)(pix * CIN + 4 * ((lane_o & 7) ^ ((row >> 2) & 7))) * 4u : S_HOOB;
      s2_dma(a_words, (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_addr + (unsigned)(buf * S_STAGE + q * 1024))), off, 0u);
    }
  };
  auto convert_row = [&](int buf, int row) {      // split the landed fp32 patch row at POSITION `row` in place (group q = k / 8 at q ^ swizzle)
    const int sw = (row >> 2) & 7;
    char* rp = smem + buf * S_STAGE + row * 128;
    float4 f[8];
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4) f[g4] = *reinterpret_cast<const float4*>(rp + ((g4 ^ sw) << 4));
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      unsigned a0, a1, a2, a3, b0, b1, b2, b3;
      s2_split_scaled(f[2 * kg].x, f[2 * kg].y, x_scale, a0, b0);
      s2_split_scaled(f[2 * kg].z, f[2 * kg].w, x_scale, a1, b1);
      s2_split_scaled(f[2 * kg + 1].x, f[2 * kg + 1].y, x_scale, a2, b2);
      s2_split_scaled(f[2 * kg + 1].z, f[2 * kg + 1].w, x_scale, a3, b3);
      u32x4s a, b;
      a.x = a0; a.y = a1; a.z = a2; a.w = a3;
      b.x = b0; b.y = b1; b.z = b2; b.w = b3;
      *reinterpret_cast<u32x4s*>(rp + ((kg ^ sw) << 4)) = a;
      *reinterpret_cast<u32x4s*>(rp + (((4 + kg) ^ sw) << 4)) = b;
    }
  };
  // rows of a patch: the two transfer waves take two rows per thread (256), the six MFMA waves the rest behind their MFMAs.  Unit u
  // -> row position: inside a group of 32 positions the lanes go 0, 1, 4, 5, 8, 9, ... then 2, 3, 6, 7, ...: sixteen lanes of a 16-byte
  // access then cover both bank halves and eight different swizzles (positions in lane order would be a 2-way conflict with the
  // (position >> 2) & 7 swizzle of this kernel)
  auto unit_pos = [&](int u) { return (u & ~31) | (4 * ((u & 15) >> 1) + (u & 1) + 2 * ((u >> 4) & 1)); };
  auto convert_share = [&](int buf) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if (t_ >= 384) {
      convert_row(buf, unit_pos(t_ - 384));
      convert_row(buf, unit_pos(t_ - 384 + 128));
    } else {
      const int pos = unit_pos(256 + t_);
      if (256 + t_ < ((9 * W + 31) & ~31) && pos < 9 * W) convert_row(buf, pos);
    }
  };

  // prologue: my first tile's patch, split
  {
    const int fp = patch_first_pixel((int)blockIdx.x);
#pragma unroll
    for (int i = 0; i < 7; ++i) issue_patch_piece(i, fp, 0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  convert_share(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();

  unsigned out_bits = 0;
  const float bias1 = p.bias1[32 * cb + fr], bias_d = p.bias_d[32 * cb + fr];
  // my pixel fr of my block: output (oy, ox) of the tile; patch row of its tap (0, 0) - (2 oy) W + 2 ox - 1 - and the taps that leave
  // the image on the left (dx = 0 at ox = 0); the top row's (dy = 0 at oy = 0) only in an image's first tile
  const int o_ = 32 * blk + fr, oy_ = o_ / Wo, ox_ = o_ - oy_ * Wo;
  const int lrow = 2 * oy_ * W + 2 * ox_ - 1;
  const unsigned mask_left = ox_ == 0 ? 0x1FFu & ~0x49u : 0x1FFu;       // taps 0, 3, 6
  const unsigned mask_top = oy_ == 0 ? 0x1FFu & ~0x7u : 0x1FFu;          // taps 0, 1, 2
  const unsigned cnt_addr = smem_addr + (unsigned)S_CNT;

#define S_PIN() __builtin_amdgcn_sched_barrier(0)
#define S_SIGNAL(ADDR)                                                                               \
  {                                                                                                  \
    unsigned long long keep_;                                                                        \
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0" \
                 : "=&s"(keep_) : "v"(ADDR), "v"(1u) : "memory");                                    \
  }
#define S_AWAIT(ADDR, TARGET)                                                                        \
  for (;;) {                                                                                         \
    unsigned seen_;                                                                                  \
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen_) : "v"(ADDR) : "memory"); \
    if ((int)(__builtin_amdgcn_readfirstlane(seen_) - (unsigned)(TARGET)) >= 0) break;               \
    __builtin_amdgcn_s_sleep(1);                                                                     \
  }
#define S_MFMA_A(ACC, TAP, S, PL, PXV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(PXV), S_AR_##TAP##_##S##_##PL(wa[S_W(TAP, S, PL)]))
#define S_MFMA_A_FIRST(ACC, TAP, S, PL, PXV) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(PXV), S_AR_##TAP##_##S##_##PL(wa[S_W(TAP, S, PL)]))
#define S_MFMA_V(ACC, WV, PXV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(PXV), "v"(WV))
#define S_MFMA_V_FIRST(ACC, WV, PXV) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(PXV), "v"(WV))
#define S_MFMA_V_LAST(ACC, WV, PXV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(ACC) : "v"(PXV), "v"(WV))
#define S_W(TAP, S, PL) (((TAP) * 4 + (S) * 2 + (PL)) & 31)
  // address of the lane's 16 bytes of tap TAP, k-step 0, first pieces - or in the block of zeros, on the same banks, when the tap
  // leaves the image.  k-step 1 is that address ^ 32, the remainder pieces ^ 64.  The lanes of a fragment read are patch rows TWO
  // apart (stride 2): patch row r sits at LDS row position (r with its two low bits swapped), its 16-byte groups XOR-swizzled by
  // (r >> 2) & 7 - sixteen lanes then cover both 128-byte bank halves and eight different groups (with rows in their natural
  // order and the stride-1 kernels' (r >> 1) & 7 swizzle every read was a 2-way conflict).
#define S_ADDR(DST, BUF, TAP)                                                                        \
  {                                                                                                  \
    const int row_ = lrow_t + ((TAP) / 3) * W + ((TAP) % 3);                                         \
    const int pos_ = (row_ & ~3) | ((row_ & 1) << 1) | ((row_ >> 1) & 1);                            \
    const unsigned a_ = (unsigned)(BUF) + (unsigned)(pos_ * 128) + (unsigned)(((fh ^ ((row_ >> 2) & 7))) << 4); \
    DST = ((rmask >> (TAP)) & 1u) ? a_ : (unsigned)S_ZROW + (a_ & 255u);                             \
  }
#define S_LOAD(DST, ADDR, S)                                                                         \
  {                                                                                                  \
    DST[0] = *reinterpret_cast<const u32x4s*>(smem + ((ADDR) ^ (32u * (S))));                        \
    DST[1] = *reinterpret_cast<const u32x4s*>(smem + ((ADDR) ^ (32u * (S)) ^ 64u));                  \
  }
  // k-step (TAP, S): three MFMAs of the 3x3 on CUR (weights' first pieces x pixels' remainders, weights' remainders x pixels' first
  // pieces, first x first), at the centre tap three more of the shortcut on the same fragments; reads the fragments of the k-step
  // AFTER NEXT - (TAP + 1, S) - into FAR (a wave has three MFMAs per k-step to cover an LDS round trip with: three register sets in
  // rotation); ADN holds the address of tap TAP + 1, ADF receives that of tap TAP + 2 (S = 1); a transfer piece behind the last.
#define S_STEP(TAP, S, CUR, FAR, PIECE, ADN, ADF)                                                    \
  {                                                                                                  \
    u32x4s w8a_, w8b_;                                                                               \
    if constexpr ((TAP) == 8) {                                                                      \
      w8a_ = *reinterpret_cast<const u32x4s*>(smem + w8 + ((S) * 2 + 0) * 1024);                     \
      w8b_ = *reinterpret_cast<const u32x4s*>(smem + w8 + ((S) * 2 + 1) * 1024);                     \
    }                                                                                                \
    if constexpr ((TAP) < 8) {                                                                       \
      if constexpr ((TAP) == 0 && (S) == 0) { S_MFMA_A_FIRST(acc, TAP, S, 0, CUR[1]); } else { S_MFMA_A(acc, TAP, S, 0, CUR[1]); } \
    } else { S_MFMA_V(acc, w8a_, CUR[1]); }                                                          \
    S_PIN();                                                                                         \
    if constexpr ((TAP) < 8) S_LOAD(FAR, ADN, S);                                                    \
    S_PIN();                                                                                         \
    if constexpr ((TAP) < 8) { S_MFMA_A(acc, TAP, S, 1, CUR[0]); } else { S_MFMA_V(acc, w8b_, CUR[0]); } \
    S_PIN();                                                                                         \
    if constexpr ((S) == 1 && (TAP) < 7) S_ADDR(ADF, rbuf, (TAP) + 2);                               \
    S_PIN();                                                                                         \
    if constexpr ((TAP) < 8) { S_MFMA_A(acc, TAP, S, 0, CUR[0]); }                                   \
    else if constexpr ((S) == 1) { S_MFMA_V_LAST(acc, w8a_, CUR[0]); } else { S_MFMA_V(acc, w8a_, CUR[0]); } \
    S_PIN();                                                                                         \
    if constexpr ((TAP) == 4) {       /* the shortcut: the centre tap is its only tap */              \
      if constexpr ((S) == 0) { S_MFMA_V_FIRST(accd, wd[0], CUR[1]); } else { S_MFMA_V(accd, wd[2], CUR[1]); } \
      S_PIN();                                                                                       \
      S_MFMA_V(accd, wd[2 * (S) + 1], CUR[0]); S_PIN();                                              \
      if constexpr ((S) == 1) { S_MFMA_V_LAST(accd, wd[2], CUR[0]); } else { S_MFMA_V(accd, wd[0], CUR[0]); } \
      S_PIN();                                                                                       \
    }                                                                                                \
    if ((PIECE) < 7 && dma_on) issue_patch_piece(PIECE, f_first, f_buf);                             \
    if ((PIECE) == 77 && dma_on) {      /* my pieces (issued 7+ k-steps ago) have landed: count them in (later - behind the loop - */ \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* the splitting waves start later: measured + 8 %) */                \
      S_SIGNAL(cnt_addr);                                                                            \
    }                                                                                                \
    S_PIN();                                                                                         \
  }

  // ---- epilogues: the 3x3 (1 / scale x sum + bias, ReLU) and the shortcut (1 / scale x sum + bias): a lane owns ONE output channel
  // (32 cb + fr) and sixteen pixels of the block (8 (r / 4) + 4 fh + r % 4): one dword store per accumulator register covers two
  // whole 128-byte half rows; two per-lane offsets plus constants in the instruction's offset field.  The second MFMA wave of a
  // SIMD (waves 4, 5) runs it BEFORE the tile's barrier, the first after it: one wave's stores under the other's MFMAs.
#define S_ROFF(OFF, R) ((((R) >> 3) ? (OFF) + 16u * COUT * 4u : (OFF)) + (unsigned)(((((R) >> 2) & 1) * 8 + ((R) & 3)) * COUT * 4))
#define S_EPILOGUE()                                                                                 \
  {                                                                                                  \
    const unsigned off = (unsigned)((tile * S_OPIX + 32 * blk + 4 * fh) * COUT + 32 * cb + fr) * 4u; \
    unsigned mx = 0;                                                                                 \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
      const float o = fmaxf(__builtin_fmaf(acc[r], unscale1, bias1), 0.f);                           \
      mx = max(mx, __float_as_uint(o));                                                              \
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), o1_rsrc, S_ROFF(off, r), 0, 0);      \
    }                                                                                                \
    out_bits = max(out_bits, mx);                                                                    \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
      const float o = __builtin_fmaf(accd[r], unscale_d, bias_d);                                    \
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), o2_rsrc, S_ROFF(off, r), 0, 0);      \
    }                                                                                                \
  }

  f32x16s acc, accd;
  for (int k = 0; k < my_tiles; ++k) {
    const int tile = blockIdx.x + k * grid;
    const unsigned rbuf = (unsigned)((k & 1) * S_STAGE);
    const bool dma_on = k + 1 < my_tiles;
    const int f_buf = (k + 1) & 1;
    const int f_first = patch_first_pixel(tile + grid);
    asm volatile("" : "+v"(lane_o));
    asm volatile("" : "+s"(wave_o));
    if (mfma_wave) {
      const int tq = tile % tiles_per_img;
      const unsigned rmask = mask_left & (tq == 0 ? mask_top : 0x1FFu);
      int lrow_t = lrow;
      asm volatile("" : "+v"(lrow_t));
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accd[e] = 0.f; }
      asm volatile("" : "+v"(acc), "+v"(accd));
      {
        u32x4s pxA[2], pxB[2], pxC[2];
        unsigned adE, adO;
        S_ADDR(adE, rbuf, 0);
        S_LOAD(pxA, adE, 0);
        S_LOAD(pxB, adE, 1);
        S_ADDR(adO, rbuf, 1);
        S_STEP(0, 0, pxA, pxC, 0, adO, adE)   S_STEP(0, 1, pxB, pxA, 1, adO, adE)
        S_STEP(1, 0, pxC, pxB, 2, adE, adO)   S_STEP(1, 1, pxA, pxC, 3, adE, adO)
        S_STEP(2, 0, pxB, pxA, 4, adO, adE)   S_STEP(2, 1, pxC, pxB, 5, adO, adE)
        S_STEP(3, 0, pxA, pxC, 6, adE, adO)   S_STEP(3, 1, pxB, pxA, 99, adE, adO)
        S_STEP(4, 0, pxC, pxB, 99, adO, adE)  S_STEP(4, 1, pxA, pxC, 99, adO, adE)
        S_STEP(5, 0, pxB, pxA, 99, adE, adO)  S_STEP(5, 1, pxC, pxB, 99, adE, adO)
        S_STEP(6, 0, pxA, pxC, 99, adO, adE)  S_STEP(6, 1, pxB, pxA, 99, adO, adE)
        S_STEP(7, 0, pxC, pxB, 77, adE, adO)  S_STEP(7, 1, pxA, pxC, 99, adE, adO)
        S_STEP(8, 0, pxB, pxA, 99, adO, adE)  S_STEP(8, 1, pxC, pxB, 99, adO, adE)
      }
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc), "+v"(accd));
      if (wave >= 4) S_EPILOGUE();
    } else if (dma_on) {
#pragma unroll
      for (int i = 0; i < 7; ++i) issue_patch_piece(i, f_first, f_buf);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      S_SIGNAL(cnt_addr);
    }
    // the next tile's patch: landed (every wave counted its pieces in), split by all; ONE barrier per tile: every wave's reads of this
    // tile's patch are done, the next patch is split
    if (dma_on) {
      S_AWAIT(cnt_addr, 8 * (k + 1))
      convert_share(f_buf);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (mfma_wave && wave < 4) S_EPILOGUE();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.out1_max) publish_abs_max(p.out1_max, out_bits);
#undef S_PIN
#undef S_SIGNAL
#undef S_AWAIT
#undef S_MFMA_A
#undef S_MFMA_A_FIRST
#undef S_MFMA_V
#undef S_MFMA_V_FIRST
#undef S_MFMA_V_LAST
#undef S_W
#undef S_ADDR
#undef S_LOAD
#undef S_STEP
#undef S_EPILOGUE
#undef S_ROFF
}

bool conv_c32s2_applicable(const Stride2Launch& c) {
  return c.in && c.out1 && c.out2 && c.w1_split && c.wd_split && c.unscale1 > 0.f && c.unscale_d > 0.f && c.in_max && c.W == 48 &&
         c.H % 8 == 0 && c.H >= 8 && c.n_img > 0 && c.num_cu > 0 && (size_t)c.n_img * c.H * c.W * 32 * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_c32s2(const Stride2Launch& c, hipStream_t s) {
  if (!conv_c32s2_applicable(c)) return hipErrorInvalidValue;
  const int n_tiles = c.n_img * (c.H / 8);
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c32s2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(conv_c32s2_kernel, dim3(grid), dim3(512), S_LDS, s, c, n_tiles);
  return hipGetLastError();
}

}  // namespace ut
