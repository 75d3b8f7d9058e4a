// 3x3 / stride-1 / 64 -> 64-channel convolution (five of layer2's six convolutions, lib/models/backbone_resnet.py:56-72 at
// 24x24x64) in the split-fp16 arithmetic of conv_split.hip, with the WEIGHTS RESIDENT IN REGISTERS.
//
// conv_split_kernel streams a 32-deep weight chunk per synchronisation and pays ~1,600 cycles per chunk on top of its MFMAs
// whatever their number: transfer issue, counted vmcnt, an LDS-counter rendezvous of its eight waves, tap
// bookkeeping.  With 64 output channels a chunk is only 12 MFMAs per wave and the kernel sits at a third of the matrix pipe.
// Here nothing is streamed but pixels:
//   * four waves per workgroup, ONE per SIMD, each with the whole 512-entry register file of its SIMD: a wave owns 32 output
//     channels and keeps their 32 x 576 weights - both fp16 planes, 72 fragments of 16 bytes per lane - in registers for the life of
//     the persistent workgroup: 64 fragments in the accumulator half of the file, 8 in the vector half;
//   * a tile is 256 consecutive pixels of the [pixel][channel] matrix; its input rows (256 + one image row and one pixel on
//     either side: <= 320 rows) come into LDS once per 32-channel slice by LDS-DMA - a ring of three slice patches, filled
//     two slice runs ahead - are split in place into the two fp16 pieces (scaled by the producer's max word, as everywhere),
//     and the nine taps read shifted rows of the patch (out-of-image taps: a block of zeros, through a per-pixel mask table);
//   * a slice run is 9 taps x 2 k-steps x (4 pixel blocks x 3 products) = 216 MFMAs per wave with no synchronisation
//     inside; two barriers per run around the in-place split of the next patch.
// One wave per SIMD issues in order, so every non-MFMA instruction sits BETWEEN MFMAs (each leaves ~24 idle issue cycles):
// fragment reads of the next k-step behind the MFMAs of this one, transfer issue a piece at a time.  The MFMAs are inline
// asm so that the operand classes are this file's choice (hipcc keeps MFMA A/B operands in the vector half and spills,
// DESIGN.md 4c); what the compiler then does not know - the wait states between an asm MFMA and other instructions that
// touch its registers - is in C64_LEAD / C64_DRAIN.
// The pixels are the MFMAs' first operand, the weights the second: a lane's accumulators are one output channel of sixteen pixels,
// so the epilogue's dword loads and stores are whole 128-byte half rows of [pixel][channel] without any exchange between lanes.
// Same interface, tensors and results as conv_split_kernel<256, 64, 8, 1, true> (bit-identical: same products, same k order
// per output element: slice, tap, k-step, product).
#include <atomic>

#include "ut_kernels.h"

namespace ut {
namespace {

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2c __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_charc;

constexpr int C_BM = 256;                 // pixels per tile
constexpr int C_HROWS = 320;              // patch rows per slice (256 + 2 * (image width + 1) <= 320: width <= 31)
constexpr int C_STAGE = C_HROWS * 128;    // one slice patch: 40 KB
constexpr int C_NBUF = 3;
constexpr int C_ZROW = C_NBUF * C_STAGE;  // 256 bytes of zeros
constexpr int C_MASK = C_ZROW + 256;      // per-pixel-of-the-image 9-bit tap validity masks (u32), up to C_MAXHW pixels
constexpr int C_MAXHW = 1024;
constexpr int C_LDS = C_MASK + C_MAXHW * 4 + 16;
constexpr int C_PIECES = C_HROWS / 8;     // 40 one-KB pieces per slice patch
constexpr int C_PW = C_PIECES / 4;        // 10 per wave
constexpr unsigned C_OOB = 0xFFFFFF00u, C_HOOB = 0x80000000u;
static_assert(C_ZROW % 256 == 0, "zero block bank-row aligned");

__device__ __forceinline__ void c_dma(u32x4c rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc), "s"(soffset)
      : "memory");
}
__device__ __forceinline__ u32x4c c_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4c r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void c_split_scaled(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2c h = __builtin_bit_cast(f16x2c, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

}  // namespace

__global__ __launch_bounds__(256, 1) void conv_c64r_kernel(ConvLaunch p, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_charc*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave & 1;                  // my 32 output channels: 32 cb .. 32 cb + 31
  const int ph = wave >> 1;                 // my pixel blocks of a tile: 2 j + ph, j = 0..3
  const int fr = lane & 31, fh = lane >> 5;
  const int wimg = p.W;
  const int hw = p.H * p.W;
  const int M = p.n_img * hw;
  constexpr int CIN = 64, COUT = 64;

  float x_scale = 1.f, x_unscale = 1.f;
  if (p.in_max) {
    bool ok;
    split_act_scale(p.in_max, nullptr, x_scale, x_unscale, ok);
    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);
  }
  const float tot_unscale = p.split_unscale * x_unscale;

  const u32x4c a_words = c_rsrc(p.in, (unsigned)((size_t)M * CIN * sizeof(float)));
  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.bias), 0, p.res ? (int)((size_t)M * COUT * sizeof(float)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)M * COUT * sizeof(float)), 0x00020000);

  // ---- my weights: group cb of ConvLaunch::w_split ([cout / 32][chunk 18][k-step 2][plane 2][lane 64][8 halves]; chunk = slice * 9 +
  // tap): fragment f = chunk * 4 + k-step * 2 + plane.  Fragments 0..63 are loaded straight into the accumulator half of the
  // register file (a value born there stays there), 64..71 (taps 7 and 8 of slice 1) into the vector half.
  u32x4c wa[64], wv[8];
  {
    const char* wg = reinterpret_cast<const char*>(p.w_split) + (size_t)cb * 18 * 4096 + lane * 16;
    // (all 64 requests first, ONE wait: sixteen dependent round trips at the head of every launch were 3 % of its time)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      asm volatile(
          "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
          "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"
          : "=&a"(wa[4 * q + 0]), "=&a"(wa[4 * q + 1]), "=&a"(wa[4 * q + 2]), "=&a"(wa[4 * q + 3])
          : "v"(wg + q * 4096)
          : "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q)      // the wait names every destination, so nothing that uses them can be scheduled above it
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+a"(wa[16 * q + 0]), "+a"(wa[16 * q + 1]), "+a"(wa[16 * q + 2]), "+a"(wa[16 * q + 3]), "+a"(wa[16 * q + 4]),
                     "+a"(wa[16 * q + 5]), "+a"(wa[16 * q + 6]), "+a"(wa[16 * q + 7]), "+a"(wa[16 * q + 8]), "+a"(wa[16 * q + 9]),
                     "+a"(wa[16 * q + 10]), "+a"(wa[16 * q + 11]), "+a"(wa[16 * q + 12]), "+a"(wa[16 * q + 13]),
                     "+a"(wa[16 * q + 14]), "+a"(wa[16 * q + 15])
                   :: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) wv[q] = *reinterpret_cast<const u32x4c*>(wg + 16 * 4096 + q * 1024);
  }

  // ---- zero block, tap-validity masks of every pixel position of an image
  if (tid < 16) *reinterpret_cast<u32x4c*>(smem + C_ZROW + tid * 16) = u32x4c{0, 0, 0, 0};
  for (int pos = tid; pos < hw; pos += 256) {
    const int y = pos / wimg, x = pos - y * wimg;
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const bool ok = (unsigned)(y + t / 3 - 1) < (unsigned)p.H && (unsigned)(x + t % 3 - 1) < (unsigned)wimg;
      mk |= (ok ? 1u : 0u) << t;
    }
    *reinterpret_cast<unsigned*>(smem + C_MASK + pos * 4) = mk;
  }

  // ---- the patch stream: run n = 2 * (my k-th tile) + slice fills buffer n % 3
  const int grid = gridDim.x;
  const int my_tiles = (n_tiles - (int)blockIdx.x + grid - 1) / grid;       // tiles blockIdx.x, + grid, ...
  const int n_runs = 2 * my_tiles;
  // (lane_o / wave_o: per-run opaque copies of lane and wave, so that the ten pieces' row / swizzle / address constants are
  // recomputed at issue time - a handful of instructions between MFMAs - instead of hoisted into registers that spill)
  int lane_o = lane, wave_o = wave;
  auto issue_patch_piece = [&](int i, int tile, int slice, int buf) {       // piece wave + 4 i of the slice patch of `tile`
    const int q = wave_o + 4 * i;
    const int row = 8 * q + (lane_o >> 3);
    const int pix = tile * C_BM - wimg - 1 + row;
    const bool ok = pix >= 0 && pix < M;
    const unsigned off = ok ? (unsigned)(pix * CIN + 4 * ((lane_o & 7) ^ ((row >> 1) & 7))) * 4u : C_HOOB;
    c_dma(a_words, (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_addr + (unsigned)(buf * C_STAGE + q * 1024))), off,
          (unsigned)slice * 128u);
  };
  auto convert_patch = [&](int buf) {       // split the landed fp32 patch in place (group q = 4 * piece + k / 8 at position q ^ swizzle)
    for (int row = tid; row < C_HROWS; row += 256) {
      const int sw = (row >> 1) & 7;
      char* rp = smem + buf * C_STAGE + row * 128;
      float4 f[8];
#pragma unroll
      for (int g4 = 0; g4 < 8; ++g4) f[g4] = *reinterpret_cast<const float4*>(rp + ((g4 ^ sw) << 4));
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        unsigned a0, a1, a2, a3, b0, b1, b2, b3;
        c_split_scaled(f[2 * kg].x, f[2 * kg].y, x_scale, a0, b0);
        c_split_scaled(f[2 * kg].z, f[2 * kg].w, x_scale, a1, b1);
        c_split_scaled(f[2 * kg + 1].x, f[2 * kg + 1].y, x_scale, a2, b2);
        c_split_scaled(f[2 * kg + 1].z, f[2 * kg + 1].w, x_scale, a3, b3);
        u32x4c a, b;
        a.x = a0; a.y = a1; a.z = a2; a.w = a3;
        b.x = b0; b.y = b1; b.z = b2; b.w = b3;
        *reinterpret_cast<u32x4c*>(rp + ((kg ^ sw) << 4)) = a;
        *reinterpret_cast<u32x4c*>(rp + (((4 + kg) ^ sw) << 4)) = b;
      }
    }
  };

  // prologue: both slices of my first tile, the first one split
#pragma unroll
  for (int i = 0; i < C_PW; ++i) issue_patch_piece(i, blockIdx.x, 0, 0);
#pragma unroll
  for (int i = 0; i < C_PW; ++i) issue_patch_piece(i, blockIdx.x, 1, 1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  convert_patch(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();

  unsigned out_bits = 0;
  int lrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lrow[j] = 32 * (2 * j + ph) + fr + wimg + 1;      // patch row of my pixel of block j, centre tap

#define C64_PIN() __builtin_amdgcn_sched_barrier(0)
#define C64_MFMA1(ACC, WCL, WV, PXV) asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+v"(ACC) : WCL(WV), "v"(PXV))
  // one pixel block's three products of a k-step: weight planes W0 (first pieces), W1 (remainders); pixel pieces PX[0], PX[1]
  // (per block and k-step three products in this order: weights' first pieces x pixels' remainders, weights' remainders x pixels'
  // first pieces, first pieces x first pieces - small terms first, like conv_split.hip)
  // Address of the lane's 16 bytes of block J, tap TAP, k-step 0, first pieces, in the patch at byte offset BUF - or in the block of
  // zeros, on the same banks, when the tap leaves the image.  k-step 1 is that address ^ 32, the remainder pieces ^ 64.
#define C64_ADDR(DST, BUF, J, TAP)                                                                   \
  {                                                                                                  \
    const int row_ = lrow_t[J] + ((TAP) / 3 - 1) * wimg + ((TAP) % 3 - 1);                           \
    const unsigned a_ = (unsigned)(BUF) + (unsigned)(row_ * 128) + (unsigned)(((fh ^ ((row_ >> 1) & 7))) << 4); \
    DST = ((rmask[J] >> (TAP)) & 1u) ? a_ : (unsigned)C_ZROW + (a_ & 255u);                          \
  }
#define C64_LOAD(DST, ADDR, S)                                                                       \
  {                                                                                                  \
    DST[0] = *reinterpret_cast<const u32x4c*>(smem + ((ADDR) ^ (32u * (S))));                        \
    DST[1] = *reinterpret_cast<const u32x4c*>(smem + ((ADDR) ^ (32u * (S)) ^ 64u));                  \
  }
#define C64_LEAD() asm volatile("s_nop 3")
#define C64_DRAIN() asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]))
  // a slice run: SLICE's weights are fragments (SLICE * 9 + tap) * 4 + S * 2 + plane: wa[..] below 64, wv[.. - 64] from there
#define C64_W(SLICE, TAP, S, PL) (((SLICE) * 9 + (TAP)) * 4 + (S) * 2 + (PL))
  // One k-step: for each of my four pixel blocks three MFMAs, with the work for the NEXT k-step of that block between them (one
  // wave per SIMD issues in order: every MFMA leaves ~24 idle issue cycles, a group of MFMAs leaves none): the address of the
  // next tap behind the first MFMA (S = 1 only: the next k-step is then a new tap), the two fragment reads behind the second,
  // a transfer piece of the patch two runs ahead behind the third (block 1 only).
#define C64_STEP(SLICE, TAP, S, CUR, NXT, HAVE_NEXT, PIECE)                                          \
  {                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
      if constexpr (C64_W(SLICE, TAP, S, 0) < 64) {                                                  \
        C64_MFMA1(acc[j], "a", wa[C64_W(SLICE, TAP, S, 0) & 63], CUR[j][1]); C64_PIN();              \
        if constexpr (HAVE_NEXT && (S) == 1) C64_ADDR(adr[j], rbuf, j, (TAP) + 1);                   \
        C64_PIN();                                                                                   \
        C64_MFMA1(acc[j], "a", wa[C64_W(SLICE, TAP, S, 1) & 63], CUR[j][0]); C64_PIN();              \
        if constexpr (HAVE_NEXT) C64_LOAD(NXT[j], adr[j], 1 - (S));                                  \
        C64_PIN();                                                                                   \
        C64_MFMA1(acc[j], "a", wa[C64_W(SLICE, TAP, S, 0) & 63], CUR[j][0]); C64_PIN();              \
      } else {                                                                                       \
        C64_MFMA1(acc[j], "v", wv[C64_W(SLICE, TAP, S, 0) & 7], CUR[j][1]); C64_PIN();               \
        if constexpr (HAVE_NEXT && (S) == 1) C64_ADDR(adr[j], rbuf, j, (TAP) + 1);                   \
        C64_PIN();                                                                                   \
        C64_MFMA1(acc[j], "v", wv[C64_W(SLICE, TAP, S, 1) & 7], CUR[j][0]); C64_PIN();               \
        if constexpr (HAVE_NEXT) C64_LOAD(NXT[j], adr[j], 1 - (S));                                  \
        C64_PIN();                                                                                   \
        C64_MFMA1(acc[j], "v", wv[C64_W(SLICE, TAP, S, 0) & 7], CUR[j][0]); C64_PIN();               \
      }                                                                                              \
      if (j == 1 && (PIECE) < C_PW && dma_on) issue_patch_piece(PIECE, f_tile, f_slice, f_buf);      \
      C64_PIN();                                                                                     \
    }                                                                                                \
  }
#define C64_RUN(SLICE)                                                                               \
  {                                                                                                  \
    u32x4c pxA[4][2], pxB[4][2];                                                                     \
    unsigned adr[4];                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { C64_ADDR(adr[j], rbuf, j, 0); C64_LOAD(pxA[j], adr[j], 0); } \
    C64_LEAD();                                                                                      \
    C64_STEP(SLICE, 0, 0, pxA, pxB, true, 0)  C64_STEP(SLICE, 0, 1, pxB, pxA, true, 1)               \
    C64_STEP(SLICE, 1, 0, pxA, pxB, true, 2)  C64_STEP(SLICE, 1, 1, pxB, pxA, true, 3)               \
    C64_STEP(SLICE, 2, 0, pxA, pxB, true, 4)  C64_STEP(SLICE, 2, 1, pxB, pxA, true, 5)               \
    C64_STEP(SLICE, 3, 0, pxA, pxB, true, 6)  C64_STEP(SLICE, 3, 1, pxB, pxA, true, 7)               \
    C64_STEP(SLICE, 4, 0, pxA, pxB, true, 8)  C64_STEP(SLICE, 4, 1, pxB, pxA, true, 9)               \
    C64_STEP(SLICE, 5, 0, pxA, pxB, true, 99) C64_STEP(SLICE, 5, 1, pxB, pxA, true, 99)              \
    C64_STEP(SLICE, 6, 0, pxA, pxB, true, 99) C64_STEP(SLICE, 6, 1, pxB, pxA, true, 99)              \
    C64_STEP(SLICE, 7, 0, pxA, pxB, true, 99) C64_STEP(SLICE, 7, 1, pxB, pxA, true, 99)              \
    C64_STEP(SLICE, 8, 0, pxA, pxB, true, 99) C64_STEP(SLICE, 8, 1, pxB, pxA, false, 99)             \
  }

  int run = 0;
  for (int k = 0; k < my_tiles; ++k) {
    const int tile = blockIdx.x + k * grid;
    // my pixels' tap masks for this tile (position in the image = pixel index modulo the image size) and opaque per-tile row
    // bases (keeps the per-tap address arithmetic inside the loop instead of hoisted into registers that live across it)
    unsigned rmask[4];
    int lrow_t[4];
    {
      const int pos0 = (tile * C_BM) % hw;      // wave-uniform (tile * 256 < M < 2^23)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = tile * C_BM + 32 * (2 * j + ph) + fr;
        int pos = pos0 + 32 * (2 * j + ph) + fr;
        pos = pos >= hw ? pos - hw : pos;
        const unsigned mk = *reinterpret_cast<const unsigned*>(smem + C_MASK + pos * 4);
        rmask[j] = m < M ? mk : 0u;
        lrow_t[j] = lrow[j];
        asm volatile("" : "+v"(lrow_t[j]));
      }
    }
    f32x16c acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[j]));

#pragma unroll
    for (int slice = 0; slice < 2; ++slice, ++run) {
      const unsigned rbuf = (unsigned)((run % C_NBUF) * C_STAGE);
      asm volatile("" : "+v"(lane_o));
      asm volatile("" : "+s"(wave_o));
      // the patch of run + 2 streams in under this run's MFMAs, into the buffer run - 1 read
      const int f_run = run + 2;
      const bool dma_on = f_run < n_runs;
      const int f_tile = blockIdx.x + (f_run >> 1) * grid, f_slice = f_run & 1, f_buf = f_run % C_NBUF;
      // second run of a tile: touch my blocks' residual lines (one dword per 128-byte line; the values are not used) so that the
      // epilogue's loads find them in L2 instead of paying an HBM round trip with nothing to overlap it
      unsigned touch[4] = {0u, 0u, 0u, 0u};
      if (slice == 1 && p.res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = tile * C_BM + 32 * (2 * j + ph) + fr;
          touch[j] = __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, m < M ? (unsigned)(m * COUT + 32 * cb) * 4u : C_OOB, 0, 0);
        }
      }
      if (slice == 0) C64_RUN(0) else C64_RUN(1)
      asm volatile("" ::"v"(touch[0]), "v"(touch[1]), "v"(touch[2]), "v"(touch[3]));
      // the patch of run + 1 (issued during run - 1) must have landed before it is split; this run's ten pieces may stay in flight
      if (dma_on) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C_PW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

      if (slice == 1) {
        // ---- epilogue: 1 / (weight scale x activation scale) x accumulator + bias + residual, ReLU, store.
        // The pixels are the MFMAs' first operand: in the accumulators a lane owns ONE output channel (32 cb + fr) and sixteen pixels
        // of the block (8 (r / 4) + 4 fh + r % 4), so that one dword access per accumulator register covers two whole 128-byte
        // half rows ([pixel][32 cb .. 32 cb + 31]): residual loads and stores are full lines with no exchange between lanes.
        C64_DRAIN();
        const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();
        const float bb = p.bias[32 * cb + fr];
        // software pipeline over the four blocks: the residual of block j + 1 is requested BEFORE block j's stores (a load behind
        // a store waits for the store: one counter, in order), so a block's loads are never younger than a store they wait for
        float rr[16], rn[16];
#define C64_PIX(J, R) (tile * C_BM + 32 * (2 * (J) + ph) + 8 * ((R) >> 2) + 4 * fh + ((R) & 3))
#define C64_RES(DST, J)                                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
          const int m_ = C64_PIX(J, r);                                                              \
          DST[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, m_ < M ? (unsigned)(m_ * COUT + 32 * cb + fr) * 4u : C_OOB, 0, 0)); \
        }
        C64_RES(rr, 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float o[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            o[r] = fmaxf(fmaf(acc[j][r], tot_unscale, bb + rr[r]), floor_v);
            const unsigned keep = C64_PIX(j, r) < M ? 0x7FFFFFFFu : 0u;
            out_bits = max(out_bits, __float_as_uint(o[r]) & keep);
          }
          if (j < 3) {
            C64_RES(rn, j + 1)
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = C64_PIX(j, r);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[r]), o_rsrc, m < M ? (unsigned)(m * COUT + 32 * cb + fr) * 4u : C_OOB, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) rr[r] = rn[r];
        }
#undef C64_RES
#undef C64_PIX
      }
      if (run + 1 < n_runs) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();          // B1: every wave's pieces of the next patch have landed; this run's reads are done
        convert_patch((run + 1) % C_NBUF);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();          // B2: the next patch is split
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.out_max) publish_abs_max(p.out_max, out_bits);
#undef C64_PIN
#undef C64_MFMA1
#undef C64_ADDR
#undef C64_LOAD
#undef C64_LEAD
#undef C64_DRAIN
#undef C64_W
#undef C64_STEP
#undef C64_RUN
}

bool conv_c64r_applicable(const ConvLaunch& c) {
  return c.w_split && c.split_unscale > 0.f && c.ksize == 3 && c.stride == 1 && c.pad == 1 && c.cin == 64 && c.cout_store == 64 &&
         c.cslice == 32 && c.k_pad == 576 && c.cout_pad >= 64 && !c.out_nchw && c.splits == 0 && c.W <= 31 && c.H == c.Ho &&
         c.W == c.Wo && c.H * c.W <= C_MAXHW && c.H * c.W >= C_BM && c.num_cu > 0 &&
         (size_t)c.n_img * c.H * c.W * 64 * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_c64r(const ConvLaunch& c, hipStream_t s) {
  if (!conv_c64r_applicable(c)) return hipErrorInvalidValue;
  const long M = (long)c.n_img * c.H * c.W;
  const int n_tiles = (int)((M + C_BM - 1) / C_BM);
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c64r_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C_LDS);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;           // one 256-thread workgroup per CU, one wave per SIMD; tiles are dealt round robin (they all cost the same)
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(conv_c64r_kernel, dim3(grid), dim3(256), C_LDS, s, c, n_tiles);
  return hipGetLastError();
}

}  // namespace ut
