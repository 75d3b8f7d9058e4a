// Split-fp16 implicit-GEMM convolution for gfx950: fp32-level results from the fp16 matrix cores.
//
// An fp32 value x is split into two fp16 pieces, x0 = fp16(x) and x1 = fp16(x - x0) (x - x0 is exact in fp32): together
// 22 significand bits, with an absolute floor of 2^-25 (half the fp16 subnormal quantum) for small values.  A product of
// two fp16 values is exact in fp32 and v_mfma_f32_32x32x16_f16 accumulates in fp32, so
//     x0 w0 + x0 w1 + x1 w0
// drops only x1 w1 and the two rounding remainders: terms of ~2^-22 of a product, against the 2^-24 of one fp32 rounding.
// At the path's outputs that is indistinguishable from a change of summation order (oracle/studies/split_precision.py:
// 2.4e-7 rad at the joint angles, 2e-4 mm at the keypoints, the same as the six-product bf16 split and as the fp32 kernels
// against the oracle), while the matrix pipe needs 3 x 32 cycles per 16 k instead of 16 x 64 / 2 = 512: 5.3x the rate
// of v_mfma_f32_32x32x2_f32.  Same layers as conv_igemm.hip (lib/models/backbone_resnet.py:56-72), same NHWC fp32
// tensors in HBM; only the arithmetic inside the kernel changes.
// Range: both operands are brought into fp16's range by powers of two, exactly.  The weights are pre-scaled per layer on
// the host so that their largest magnitude sits just under 2^15; the activations are multiplied, before the split, by the
// power of two that puts the LAYER's largest activation in [2^14, 2^15) - taken from a device word the producing kernel
// left (ut_kernels.h::split_act_scale / publish_abs_max), so no host synchronisation.  The first piece then never
// saturates and the second piece of every value within 2^-18 of the largest is a normal fp16 number; smaller ones keep an
// absolute error of 2^-40 of the layer's largest activation.  The epilogue multiplies by the inverse of both scales.  The
// split therefore has fp32's exponent range: a network whose activations are 2^-16 or 2^+16 of this one's gives the same
// bits (tests/test_gpu_parity.py::test_split_f16_power_of_two_rescale_invariance), like the fp32 kernels.
//
// Activations stay fp32 in HBM and in LDS and are split in registers after the fragment read (6 VALU per pair of
// values: v_cvt_pkrtz_f16_f32, two conversions back, two subtractions, v_cvt_pkrtz_f16_f32); the weights' pieces are made
// once on the host and stored as fp16 planes in fragment order, so a wave reads a weight fragment with one
// conflict-free ds_read_b128 per (32 rows, 16 k, plane) - and they take the same 4 bytes per value as fp32.
//
// One workgroup of 8 waves per CU computes 256 x BN output tiles, K in chunks of 32 (two MFMA k-steps).  Operands go
// global -> LDS by LDS-DMA.  Weights: a ring of three 1-chunk stages.  Pixels: either the im2col gather of conv_igemm.hip
// into a ring of three stages (XOR-swizzled 128-byte rows), or - HALO, the stride-1 convolutions - the tile's input rows
// of a 32-channel slice brought in once per slice (two buffers) and read by the nine taps at shifted rows.
// One synchronisation per chunk through two LDS counters, each signalled long before it is awaited and looked at a few
// MFMA slots before its use: LANDED (my pieces of the next chunk are in LDS: a counted vmcnt that leaves the youngest
// batch in flight) before the reads of the next chunk, READ (my fragment reads of this chunk are done) before the
// transfers into the buffers the chunk frees.  The transfers are spread over the chunk's second k-step.
// Persistent grid with the same tile queue as conv_igemm.hip; the fetch side runs across tile boundaries.
// Epilogue: scale + bias + residual + ReLU + 16-byte NHWC stores (the accumulator layout of the 32x32 MFMAs is the same
// as in conv_igemm.hip: a lane owns one pixel, register quads are 4 consecutive channels).
#include <math.h>

#include <atomic>

#include "ut_kernels.h"

namespace ut {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) unsigned lds_u32;

constexpr unsigned OOB = 0xFFFFFF00u;

// LDS-DMA piece (see conv_igemm.hip::dma16): per-lane byte offset + wave-uniform byte offset
__device__ __forceinline__ void dma_piece(u32x4 rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc), "s"(soffset)
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
__device__ __forceinline__ int fdiv(int n, int d, float inv_d) {
  int q = (int)((float)n * inv_d);
  int r = n - q * d;
  if (r < 0) --q;
  if (r >= d) ++q;
  return q;
}

// two fp32 values -> their fp16 pieces (first, remainder), each packed {b, a}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1) {
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
// the pieces of a * s and b * s for a power of two s: the products are exact, so fma(a, s, -h) is the remainder (a * s) - h
// in one instruction that also converts h (v_fma_mix_f32)
__device__ __forceinline__ void split_pair_scaled(float a, float b, float s, unsigned& p0, unsigned& p1) {
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
  const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
__device__ __forceinline__ f16x8 frag(const unsigned (&v)[4]) {
  u32x4 t;
  t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
  return __builtin_bit_cast(f16x8, t);
}

// HALO (stride-1 3x3 convolutions from 64 input channels): the pixels of a 32-channel slice are not gathered tap by tap;
// the tile's input rows - 256 consecutive pixels plus one image row and one pixel on either side, a CONTIGUOUS range of
// the [pixel][channel] matrix - are brought into LDS once per slice (two buffers) and the nine taps read shifted rows of
// that patch, with a per-pixel 9-bit validity mask pointing out-of-image taps at a row of zeros.  Per chunk the
// workgroup then fetches 4.4 KB of pixels instead of 32.
template <int BM, int BN, int WR, int WC, bool HALO>
__global__ __launch_bounds__(512, 1) void conv_split_kernel(ConvLaunch p, int tiles_n, int n_tiles) {
  static_assert(WR * WC == 8, "8 waves per workgroup");
  constexpr int MI = BM / WR / 32, NI = BN / WC / 32;
  constexpr int AP = BM / 64;              // pixel pieces per wave per chunk (a piece = 8 rows x 128 B)
  constexpr int WTOT = BN / 32 * 4;        // 1-KB weight blocks per chunk: (32 rows) x (k-step) x (plane)
  constexpr int WP = (WTOT + 7) / 8;       // ... per wave
  constexpr int HROWS = 320;               // HALO: patch rows per slice (256 + 2 * (image width + 1) <= 320: width <= 31)
  constexpr int NPH = HROWS / 64;          // ... 1-KB pieces per wave
  constexpr int A_STAGE = HALO ? HROWS * 128 : BM * 128, W_STAGE = BN * 128;
  constexpr int A_RING = HALO ? 2 : 3, W_RING = HALO ? 4 : 3;   // stages; the weight stream runs 3 chunks ahead
  constexpr int W_AHEAD = 3;
  constexpr int W_BASE = A_RING * A_STAGE;
  constexpr int ZROW = W_BASE + W_RING * W_STAGE;          // 256 bytes of zeros (HALO: the target of out-of-image taps): a masked
                                                           // lane reads at its valid address modulo 256, i.e. on the banks the
                                                           // swizzle gave it - one fixed zero row made every masked lane collide
                                                           // with whichever valid lane owned those banks (17 % of the LDS cycles)
  static_assert(ZROW % 256 == 0, "the zero block is bank-row aligned");
  constexpr int SLOT = ZROW + 256;
  constexpr unsigned HOOB = 0x80000000u;   // out-of-range offset that stays out of range with a slice offset added
  constexpr int NM = 3 * MI * NI;          // MFMAs (= slots) per k-step; a chunk has 2 * NM
  constexpr int UNITS = MI * 4;            // pair conversions per k-step, one per slot from CV0
  constexpr int NWF = NI * 2;              // weight-fragment reads per k-step, one per slot from RD0
  constexpr int RD0 = NM >= 12 ? 2 : 0;
  constexpr int CV0 = NM >= 12 ? 4 : 2;
  // Chunk synchronisation through two LDS counters, each signalled long before it is waited for (with one workgroup
  // per CU nothing hides an s_barrier's turn-around; with the two halves 1-2 slots apart it still cost 18 %):
  //   LANDED  signalled at slot 0 of the chunk: my pieces of the NEXT chunk are in LDS (counted vmcnt);
  //           awaited before the first read of the next chunk (start of the second k-step).
  //   READ    signalled at slot ARR: my fragment reads of THIS chunk are done;
  //           awaited at chunk slot RW, in front of my first transfer into the buffers this chunk frees.
  constexpr int ARR = RD0 + NWF + (NM >= 12 ? 2 : 0);   // behind the chunk's last fragment read and in front of the first look at a
                                                         // counter: its lgkmcnt(0) then waits for fragment reads issued slots ago only
  constexpr int RW = NM + (NM >= 12 ? 1 : 0);            // chunk slot (second k-step) of the READ wait
  constexpr int PS0 = RW + 1;                            // chunk slot of the wave's first transfer
  constexpr int PEEK = NM >= 12 ? 3 : 2;                 // slots between the look at a counter and its use
  static_assert(RW - PEEK >= ARR && NM - PEEK >= ARR, "the counters are looked at after the wave's READ signal");
  constexpr int NPC = HALO ? WP + 1 : WP + AP;           // transfers of a wave per chunk
  constexpr int PSTEP = (2 * NM - PS0) / NPC;            // slots between two transfers of a wave
  constexpr int ASEP = PSTEP >= 2 ? 1 : 0;               // a pixel piece's offset arithmetic one slot ahead of its transfer
  static_assert(ARR < NM && CV0 + UNITS <= NM, "READ signal and conversions inside the first k-step");
  static_assert(PSTEP >= 1 && PS0 + (NPC - 1) * PSTEP + ASEP < 2 * NM, "room for the transfers");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WC, wn = wave % WC;
  const int g = tid & 7;
  const int r0 = tid >> 3;                  // 0..63: pixel row this thread stages (then +64 per piece)
  const int gk = g ^ ((r0 >> 1) & 7);       // which 16-byte group of the 128-byte k chunk lands at position g
  const int fr = lane & 31, fh = lane >> 5;

  const int M = p.n_img * p.Ho * p.Wo;
  const int hw = p.Ho * p.Wo;
  const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)p.Wo;
  const int taps = p.ksize * p.ksize;
  const int n_chunks = p.k_pad / 32;

  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.bias), 0, p.res ? (int)((size_t)M * p.cout_store * sizeof(float)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)M * p.cout_store * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.tile_counter, 0, 4, 0x00020000);
  const unsigned q_off = tid == 0 ? 0u : OOB;
  const u32x4 a_words = rsrc_words(p.in, (unsigned)((size_t)p.n_img * p.H * p.W * p.cin * sizeof(float)));
  const u32x4 w_words = rsrc_words(p.w_split, (unsigned)((size_t)(p.cout_pad / 32) * n_chunks * 4096));
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_char*)smem;

  // power-of-two activation scale from the producer's max word, and the epilogue's factor 2^-(weight scale + activation scale)
  float x_scale = 1.f, x_unscale = 1.f;
  if (p.in_max) {
    bool ok;
    split_act_scale(p.in_max, p.in_obs, x_scale, x_unscale, ok);
    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);
  }
  const float tot_unscale = p.split_unscale * x_unscale;

  const int grid = gridDim.x;
  int slot = blockIdx.x;
  if ((grid & 7) == 0) slot = (blockIdx.x & 7) * (grid >> 3) + (blockIdx.x >> 3);

  // ---- fetch side: the pixel stream runs three chunks ahead of the MFMAs, the weight stream two
  int fa_tile = slot, fa_tap = 0, fa_chb = 0;      // pixel stream: tile, tap, channel-slice base
  unsigned fa_stage = 0;                                      // byte offset of the ring stage it writes next
  int a_pix[AP], a_iy[AP], a_ix[AP];
  int fw_tile = slot, fw_c = 0;
  unsigned fw_stage = 0;
  unsigned fw_row;                                            // byte offset of (tile column, chunk 0) of this wave's first block
  const unsigned w_lane = (unsigned)lane * 16u;
  int next_tile = 0;                                          // the tile after the one being computed (valid once read)

  // ---- HALO: patch stream (one slice ahead of the MFMAs) and the read side's row arithmetic
  const int n_slices = p.cin / 32;
  const int wimg = p.W;
  unsigned h_off[NPH];          // per-lane byte offset of this wave's pieces of the fetch tile's patch (slice 0)
  unsigned h_buf = 0;           // byte offset of the patch buffer being filled
  int c_tap = 0, c_slice = 0;   // tap and slice of the chunk being computed
  int r_tap = 0;                // tap of the chunk the fragment reads are at (one k-step ahead of the MFMAs)
  unsigned r_buf = 0;
  unsigned rmask[MI];           // 9 validity bits (tap order) of the lane's pixel in fragment i of the tile being read
  int lrow[MI];                 // its row in the patch for the centre tap
#pragma unroll
  for (int i = 0; i < MI; ++i) { lrow[i] = wm * (MI * 32) + i * 32 + fr + wimg + 1; rmask[i] = 0; }
#define SP_H_SETUP(TILE)                                                                             \
  {                                                                                                  \
    const int g0_ = ((TILE) / tiles_n) * BM - wimg - 1;                                              \
    _Pragma("unroll") for (int q = 0; q < NPH; ++q) {                                                \
      const int row_ = 8 * (wave + 8 * q) + (lane >> 3);                                             \
      const int pix_ = g0_ + row_;                                                                   \
      const bool ok_ = pix_ >= 0 && pix_ < M && (TILE) < n_tiles;                                    \
      h_off[q] = ok_ ? (unsigned)(pix_ * p.cin + 4 * ((lane & 7) ^ ((row_ >> 1) & 7))) * 4u : HOOB;  \
    }                                                                                                \
  }
#define SP_H_ISSUE(Q, SLICE) dma_piece(a_words, smem_addr + h_buf + (unsigned)((wave + 8 * (Q)) * 1024), h_off[Q], (unsigned)(SLICE) * 128u)
  // The patch in buffer BUF, landed as fp32 rows (16-byte group g of a row at position g ^ swizzle), is split IN PLACE, once:
  // a value's two fp16 pieces take its 4 bytes.  Lane l < 40 of every wave owns row 40 * wave + l: 8 reads, 16 pair
  // conversions, 8 writes - group q = 4 * piece + k / 8 at position q ^ swizzle.  Every tap (and both waves that share the
  // rows) then reads ready pieces: the split costs 1/18 of what it costs behind every fragment read.
#define SP_H_CONVERT(BUF)                                                                            \
  if (lane < HROWS / 8) {                                                                            \
    const int row_ = (HROWS / 8) * wave + lane;                                                      \
    const int sw_ = (row_ >> 1) & 7;                                                                 \
    char* rp_ = smem + (BUF) + row_ * 128;                                                           \
    float4 f_[8];                                                                                    \
    _Pragma("unroll") for (int g4 = 0; g4 < 8; ++g4) f_[g4] = *reinterpret_cast<const float4*>(rp_ + ((g4 ^ sw_) << 4)); \
    _Pragma("unroll") for (int kg = 0; kg < 4; ++kg) {                                               \
      unsigned a0_, a1_, a2_, a3_, b0_, b1_, b2_, b3_;                                               \
      split_pair_scaled(f_[2 * kg].x, f_[2 * kg].y, x_scale, a0_, b0_);                              \
      split_pair_scaled(f_[2 * kg].z, f_[2 * kg].w, x_scale, a1_, b1_);                              \
      split_pair_scaled(f_[2 * kg + 1].x, f_[2 * kg + 1].y, x_scale, a2_, b2_);                      \
      split_pair_scaled(f_[2 * kg + 1].z, f_[2 * kg + 1].w, x_scale, a3_, b3_);                      \
      u32x4 a_, b_;                                                                                  \
      a_.x = a0_; a_.y = a1_; a_.z = a2_; a_.w = a3_;                                                \
      b_.x = b0_; b_.y = b1_; b_.z = b2_; b_.w = b3_;                                                \
      *reinterpret_cast<u32x4*>(rp_ + ((kg ^ sw_) << 4)) = a_;                                       \
      *reinterpret_cast<u32x4*>(rp_ + (((4 + kg) ^ sw_) << 4)) = b_;                                 \
    }                                                                                                \
  }
#define SP_H_MASK(TILE)                                                                              \
  {                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                 \
      const int m_ = ((TILE) / tiles_n) * BM + wm * (MI * 32) + i * 32 + fr;                         \
      const bool in_ = m_ < M && (TILE) < n_tiles;                                                   \
      const int mm_ = in_ ? m_ : 0;                                                                  \
      const int img_ = fdiv(mm_, hw, inv_hw);                                                        \
      const int rem_ = mm_ - img_ * hw;                                                              \
      const int y_ = fdiv(rem_, wimg, inv_wo), x_ = rem_ - y_ * wimg;                                \
      unsigned mk_ = 0;                                                                              \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                \
        const bool ok_ = (unsigned)(y_ + t / 3 - 1) < (unsigned)p.H && (unsigned)(x_ + t % 3 - 1) < (unsigned)wimg; \
        mk_ |= (ok_ ? 1u : 0u) << t;                                                                 \
      }                                                                                              \
      rmask[i] = in_ ? mk_ : 0u;                                                                     \
    }                                                                                                \
  }

#define SP_A_SETUP()                                                                                 \
  {                                                                                                  \
    const int tm_ = fa_tile / tiles_n;                                                               \
    _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                                 \
      const int m = tm_ * BM + r0 + 64 * i;                                                          \
      const bool ok = m < M && fa_tile < n_tiles;                                                    \
      const int mm = ok ? m : 0;                                                                     \
      const int img = fdiv(mm, hw, inv_hw);                                                          \
      const int rem = mm - img * hw;                                                                 \
      const int oy = fdiv(rem, p.Wo, inv_wo), ox = rem - oy * p.Wo;                                  \
      a_iy[i] = ok ? oy * p.stride - p.pad : -100000;                                                \
      a_ix[i] = ox * p.stride - p.pad;                                                               \
      a_pix[i] = ((img * p.H + a_iy[i]) * p.W + a_ix[i]) * p.cin + 4 * gk;                           \
    }                                                                                                \
    fa_tap = 0; fa_chb = 0;                                                                \
  }
#define SP_W_SETUP()                                                                                 \
  {                                                                                                  \
    const int tn_ = fw_tile - (fw_tile / tiles_n) * tiles_n;                                         \
    fw_row = (unsigned)(tn_ * (BN / 32)) * (unsigned)n_chunks * 4096u;                               \
    fw_c = 0;                                                                                        \
  }
  // one pixel piece of the fetch chunk: per-lane offset, then the transfer
  int f_dy = 0, f_dx = 0, f_tap_off = 0;   // tap of the pixel stream's chunk (wave-uniform)
#define SP_A_TAP()                                                                                   \
  {                                                                                                  \
    f_dy = 0; f_dx = 0;                                                                              \
    if (p.ksize == 3) { f_dy = (fa_tap * 11) >> 5; f_dx = fa_tap - 3 * f_dy; }                       \
    f_tap_off = (f_dy * p.W + f_dx) * p.cin + fa_chb;                                                \
  }
#define SP_A_ADDR(I, OFF)                                                                            \
  {                                                                                                  \
    const int iy = a_iy[I] + f_dy, ix = a_ix[I] + f_dx;                                              \
    const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;                    \
    OFF = ok ? (unsigned)(a_pix[I] + f_tap_off) * 4u : OOB;                                          \
  }
#define SP_A_ISSUE(I, OFF) dma_piece(a_words, smem_addr + fa_stage + (unsigned)((64 * (I) + 8 * wave) * 128), OFF, 0u)
#define SP_A_ADVANCE()                                                                               \
  {                                                                                                  \
    fa_stage = fa_stage + A_STAGE == A_RING * A_STAGE ? 0u : fa_stage + A_STAGE;                     \
    ++fa_tap;                                                                                \
    if (fa_tap >= taps) { fa_tap = 0; fa_chb += 32; }                                                \
  }
  // weight block T of this wave: block q = wave * WP + T of the chunk image; group q / 4 of the tile column
#define SP_W_ISSUE(T)                                                                                \
  {                                                                                                  \
    const int q_ = wave * WP + (T);                                                                  \
    const unsigned src_ = fw_row + (unsigned)((q_ / 4) * n_chunks + fw_c) * 4096u + (unsigned)(q_ % 4) * 1024u; \
    dma_piece(w_words, smem_addr + W_BASE + fw_stage + (unsigned)q_ * 1024u, w_lane, src_);          \
  }
#define SP_W_ADVANCE() { fw_stage = fw_stage + W_STAGE == W_RING * W_STAGE ? 0u : fw_stage + W_STAGE; ++fw_c; }

  // ---- compute side
  f32x16 acc[MI][NI];
  unsigned xp[2][MI][2][4];      // pixel fragments, pieces (first, remainder), two k-steps in flight
  unsigned wf[2][NI][2][4];      // weight fragments
  float4 xr[MI][2];              // raw fp32 pixels of the k-step being converted
  unsigned rd_a = 0, rd_w = 0;   // ring stages being read
  // per-lane read offsets: pixel row fr of the wave's rows, 16-byte positions (4s + 2fh + h) ^ swizzle
  const unsigned x_row = (unsigned)((wm * MI * 32 + fr) * 128);
  unsigned x_pos[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) x_pos[s][h] = (unsigned)(((4 * s + 2 * fh + h) ^ ((fr >> 1) & 7)) * 16);
  const unsigned w_rd = (unsigned)(W_BASE + wn * NI * 4 * 1024) + w_lane;

#define SP_READ_W1(SET, S, IDX)                                                                      \
  {                                                                                                  \
    constexpr int j_ = (IDX) / 2, pl_ = (IDX) % 2;                                                   \
    const u32x4 t_ = *reinterpret_cast<const u32x4*>(smem + w_rd + rd_w + ((j_ * 2 + (S)) * 2 + pl_) * 1024); \
    wf[SET][j_][pl_][0] = t_.x; wf[SET][j_][pl_][1] = t_.y; wf[SET][j_][pl_][2] = t_.z; wf[SET][j_][pl_][3] = t_.w; \
  }
#define SP_READ_X(S, A_ST)                                                                           \
  {                                                                                                  \
    if constexpr (HALO) {     /* row of the patch for tap r_tap, or the row of zeros */              \
      const int dy_ = (r_tap * 11) >> 5, dx_ = r_tap - 3 * dy_;                                      \
      const int sh_ = (dy_ - 1) * wimg + dx_ - 1;                                                    \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                               \
        const int row_ = lrow[i] + sh_;                                                              \
        /* the patch is already split (SP_H_CONVERT): 16-byte group 2S + fh of the first pieces, +4 for the remainders */ \
        const unsigned a_ = r_buf + (unsigned)(row_ * 128) + (unsigned)((((2 * (S) + fh) ^ ((row_ >> 1) & 7))) << 4); \
        const unsigned a0_ = ((rmask[i] >> r_tap) & 1u) ? a_ : (unsigned)ZROW + (a_ & 255u);         \
        const u32x4 p0_ = *reinterpret_cast<const u32x4*>(smem + a0_);                               \
        const u32x4 p1_ = *reinterpret_cast<const u32x4*>(smem + (a0_ ^ 64u));                       \
        xp[S][i][0][0] = p0_.x; xp[S][i][0][1] = p0_.y; xp[S][i][0][2] = p0_.z; xp[S][i][0][3] = p0_.w; \
        xp[S][i][1][0] = p1_.x; xp[S][i][1][1] = p1_.y; xp[S][i][1][2] = p1_.z; xp[S][i][1][3] = p1_.w; \
      }                                                                                              \
    } else {                                                                                         \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                 \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                \
          xr[i][h] = *reinterpret_cast<const float4*>(smem + (A_ST) + x_row + i * 4096 + x_pos[S][h]); \
    }                                                                                                \
  }
#define SP_READ_WALL(SET, S)                                                                         \
  {                                                                                                  \
    SP_READ_W1(SET, S, 0) SP_READ_W1(SET, S, 1)                                                      \
    if constexpr (NWF > 2) { SP_READ_W1(SET, S, 2) SP_READ_W1(SET, S, 3) }                           \
  }
  // conversion unit U of a k-step: pair (U & 3) of row fragment U >> 2
#define SP_CONV(SET, U)                                                                              \
  {                                                                                                  \
    constexpr int i_ = (U) >> 2, pr_ = (U) & 3;                                                      \
    const float4 v_ = xr[i_][pr_ >> 1];                                                              \
    if constexpr ((pr_ & 1) == 0) split_pair_scaled(v_.x, v_.y, x_scale, xp[SET][i_][0][pr_], xp[SET][i_][1][pr_]); \
    else split_pair_scaled(v_.z, v_.w, x_scale, xp[SET][i_][0][pr_], xp[SET][i_][1][pr_]);          \
  }
#define SP_PIN() __builtin_amdgcn_sched_barrier(0)
  // MFMA N of a k-step: products (weight piece, pixel piece) small terms first, accumulators round-robin inside a product
#define SP_MFMA(SET, N)                                                                              \
  {                                                                                                  \
    constexpr int pr_ = (N) / (MI * NI), ij_ = (N) % (MI * NI), i_ = ij_ / NI, j_ = ij_ % NI;        \
    constexpr int wp_ = pr_ == 0 ? 0 : pr_ == 1 ? 1 : 0;                                             \
    constexpr int xp_ = pr_ == 0 ? 1 : 0;                                                            \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(xp[SET][i_][xp_]), frag(wf[SET][j_][wp_]), acc[i_][j_], 0, 0, 0); \
  }

#define SP_SIGNAL(ADDR, LGKM)                                                                         \
  {                                                                                                  \
    unsigned long long keep_;                                                                        \
    if constexpr (LGKM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           \
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0" \
                 : "=&s"(keep_) : "v"(ADDR), "v"(1u) : "memory");                                    \
  }
  /* a counter is LOOKED AT a few MFMA slots before it is needed (a plain LDS load: the compiler places the wait at its
     first use), so that the common case costs no LDS round trip in front of the MFMAs; only a value that is still short
     falls into the polling loop */                                                                   \
#define SP_PEEK(ADDR) (*reinterpret_cast<volatile lds_u32*>(ADDR))
#define SP_AWAIT(PEEKED, ADDR, TARGET)                                                               \
  if ((int)(__builtin_amdgcn_readfirstlane(PEEKED) - (TARGET)) < 0) {                                \
    for (;;) {                                                                                       \
      unsigned seen_;                                                                                \
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen_) : "v"(ADDR) : "memory"); \
      if ((int)(__builtin_amdgcn_readfirstlane(seen_) - (TARGET)) >= 0) break;                       \
    }                                                                                                \
  }
  // One MFMA slot of a k-step.  CUR: register set consumed, NXT: set filled for the following step.  FIRST: the chunk's
  // first step, which carries the arrival and the wait of the chunk's synchronisation.
#define SP_SLOT(CUR, NXT, N, FIRST)                                                                  \
  {                                                                                                  \
    if constexpr (FIRST && (N) == 0) {                                                               \
      /* LANDED: my pieces of the next chunk are in LDS (the youngest batch may stay in flight).  The two chunks after \
         an epilogue need no wait: the epilogue's residual loads were awaited in issue order, i.e. behind every      \
         transfer issued before them, and a counted wait here would also wait for the tile's stores (vmcnt counts    \
         them, and they complete out of order with the transfers) */                                  \
      if (skip_waits > 0) --skip_waits;                                                              \
      else if constexpr (!HALO) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP + AP) : "memory");       \
      else if (prev_had_patch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP + 1) : "memory");         \
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP) : "memory");                                 \
      SP_SIGNAL(landed_addr, 0);                                                                              \
    }                                                                                                \
    /* READ is needed only where a stage is refilled in the chunk that read it (rings of 3: the gather instantiation). \
       With the HALO instantiation's weight ring of 4 the stage refilled in chunk g is that of chunk g - 1, and every   \
       wave that has signalled LANDED in chunk g (awaited before the second k-step) has finished reading it; the same   \
       holds for the patch buffer of the previous slice.  One synchronisation per chunk instead of two. */ \
    if constexpr (!HALO && FIRST && (N) == ARR) SP_SIGNAL(read_addr, 1);    /* lgkmcnt(0) first */      \
    if constexpr (!HALO && (FIRST ? 0 : NM) + (N) == RW - PEEK) read_seen = SP_PEEK(read_addr);       \
    if constexpr (FIRST && (N) == NM - PEEK) landed_seen = SP_PEEK(landed_addr);                     \
    if constexpr (!FIRST && NM + (N) == RW) {                                                        \
      if constexpr (!HALO) {                                                                         \
        read_target += 8u;                                                                           \
        SP_AWAIT(read_seen, read_addr, read_target);                                                 \
      }                                                                                              \
      /* the ticket taken at the tile's start is older than every transfer still in flight here */    \
      if (c == 1 && tid == 0) {                                                                      \
        int t_ = ticket;                                                                             \
        asm volatile("" : "+v"(t_));   /* first use HERE: hoisted above the loop it would wait (vmcnt) at the tile's start */ \
        asm volatile("ds_write_b32 %0, %1" ::"v"(slot_addr), "v"(grid + t_) : "memory");             \
      }                                                                                              \
    }                                                                                                \
    /* transfers: piece k of the wave (weights first) at chunk slot PS0 + k * PSTEP */                \
    constexpr int g_ = (FIRST ? 0 : NM) + (N) - PS0;                                                 \
    if constexpr (g_ >= 0 && g_ % PSTEP == 0 && g_ / PSTEP < WP) SP_W_ISSUE(g_ / PSTEP);              \
    if constexpr (!HALO && g_ >= 0 && g_ % PSTEP == 0 && g_ / PSTEP >= WP && g_ / PSTEP < WP + AP) SP_A_ADDR(g_ / PSTEP - WP, a_off); \
    if constexpr (!HALO && g_ >= ASEP && (g_ - ASEP) % PSTEP == 0 && (g_ - ASEP) / PSTEP >= WP && (g_ - ASEP) / PSTEP < WP + AP) \
      SP_A_ISSUE((g_ - ASEP) / PSTEP - WP, a_off);                                                   \
    if constexpr (HALO && g_ == WP * PSTEP) {      /* one piece of the next slice's patch in the slice's first NPH chunks */ \
      if (c_tap == 0) SP_H_ISSUE(0, f_slice);                                                        \
      else if (c_tap == 1) SP_H_ISSUE(1, f_slice);                                                   \
      else if (c_tap == 2) SP_H_ISSUE(2, f_slice);                                                   \
      else if (c_tap == 3) SP_H_ISSUE(3, f_slice);                                                   \
      else if (c_tap == 4) SP_H_ISSUE(4, f_slice);                                                   \
    }                                                                                                \
    if constexpr (!HALO && (N) >= CV0 && (N) - CV0 < UNITS) SP_CONV(NXT, (N) - CV0);                 \
    if constexpr ((N) >= RD0 && (N) - RD0 < NWF) SP_READ_W1(NXT, FIRST ? 1 : 0, (N) - RD0);          \
    SP_PIN();                                                                                        \
    SP_MFMA(CUR, N);                                                                                 \
    SP_PIN();                                                                                        \
  }
#define SP_SLOTS3(CUR, NXT, N, FIRST) SP_SLOT(CUR, NXT, N, FIRST) SP_SLOT(CUR, NXT, (N) + 1, FIRST) SP_SLOT(CUR, NXT, (N) + 2, FIRST)
#define SP_STEP(CUR, NXT, FIRST)                                                                     \
  {                                                                                                  \
    __builtin_amdgcn_s_setprio(2);                                                                   \
    SP_SLOTS3(CUR, NXT, 0, FIRST) SP_SLOTS3(CUR, NXT, 3, FIRST)                                      \
    if constexpr (NM > 6) { SP_SLOTS3(CUR, NXT, 6, FIRST) SP_SLOTS3(CUR, NXT, 9, FIRST) }            \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  }
  static_assert(NM == 6 || NM == 12, "SP_STEP expands 6 or 12 slots");
  static_assert(NPH == 5, "SP_SLOT issues patch pieces 0..4");

  const unsigned slot_addr = smem_addr + (unsigned)SLOT;
  const unsigned read_addr = slot_addr + 4, landed_addr = slot_addr + 8;     // the two counters (monotonic)
  unsigned read_target = 0, landed_target = 0, read_seen = 0, landed_seen = 0;
  if (tid == 0) asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %2, %1" ::"v"(read_addr), "v"(0u), "v"(landed_addr) : "memory");
  // ---- prologue: chunks 0, 1 and 2 of the first tile (HALO: the patch of its first slice and three weight chunks)
  if constexpr (!HALO) SP_A_SETUP();
  SP_W_SETUP();
  unsigned a_off = 0;
  int skip_waits = 0;
  bool prev_had_patch = false;
  int f_slice = 0;              // HALO: slice of the patch being fetched
  if (lane < 16 && wave == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(smem_addr + (unsigned)ZROW + (unsigned)lane * 16u), "v"(u32x4{0, 0, 0, 0}) : "memory");
  if constexpr (HALO) {
    SP_H_SETUP(slot);
    SP_H_MASK(slot);
    SP_H_ISSUE(0, 0); SP_H_ISSUE(1, 0); SP_H_ISSUE(2, 0); SP_H_ISSUE(3, 0); SP_H_ISSUE(4, 0);
    h_buf = A_STAGE;
    f_slice = n_slices > 1 ? 1 : 0;
  }
#pragma unroll
  for (int c = 0; c < W_AHEAD; ++c) {
    if constexpr (!HALO) {
      SP_A_TAP();
#pragma unroll
      for (int i = 0; i < AP; ++i) { SP_A_ADDR(i, a_off); SP_A_ISSUE(i, a_off); }
      SP_A_ADVANCE();
    }
#pragma unroll
    for (int t = 0; t < WP; ++t) SP_W_ISSUE(t);
    SP_W_ADVANCE();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if constexpr (HALO) {
    SP_H_CONVERT(0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  SP_READ_X(0, 0u);
  SP_READ_WALL(0, 0);
  if constexpr (!HALO) {
    SP_CONV(0, 0) SP_CONV(0, 1) SP_CONV(0, 2) SP_CONV(0, 3)
    if constexpr (UNITS > 4) { SP_CONV(0, 4) SP_CONV(0, 5) SP_CONV(0, 6) SP_CONV(0, 7) }
  }

  // The residual of a tile: requested in front of the stores (see the epilogue); with 32-row waves (half the accumulators)
  // there are registers to request it a whole chunk ahead, under the tile's last MFMAs.
  constexpr bool EARLY_RES = MI * NI <= 2;
  // The pixels are the MFMAs' FIRST operand: a lane's sixteen accumulator registers of a 32 x 32 block are ONE output channel (fr) of
  // sixteen pixels (8 (r / 4) + 4 fh + r % 4), so one dword access per register covers two whole 128-byte half rows of the
  // [pixel][channel] tensor (with the weights first a lane owned a pixel and 16 of its channels: 16-byte accesses scattered over
  // 32 rows per instruction, and the epilogue's loads and stores were a sixth of a 128-channel layer's time).
  // Block (i, j), register r: pixel tm BM + SP_PIX(i, r), channel tn BN + (wn NI + j) 32 + fr.  A block's sixteen offsets are one
  // per-lane base plus multiples of the pixel stride, walked with one add per access; pixels beyond the tensor (last tile) are beyond
  // the descriptors' range - loads return zero, stores are dropped, no compare per access - and a channel beyond it (never, for the
  // backbone's 64 / 128 / 256) starts from an offset that stays out of range under every such multiple.
#define SP_PIX(I, R) (wm * (MI * 32) + (I) * 32 + 8 * ((R) >> 2) + 4 * fh + ((R) & 3))
#define SP_BASE(TM, TN, J) ((TN) * BN + wn * (NI * 32) + (J) * 32 + fr < p.cout_store                   \
                                ? (unsigned)(((TM) * BM + wm * (MI * 32) + 4 * fh) * p.cout_store + (TN) * BN + wn * (NI * 32) + (J) * 32 + fr) * 4u \
                                : HOOB)
  float rr[MI][NI][16];
#define SP_RES_REQUEST()                                                                             \
  {                                                                                                  \
    const int tm_ = tile / tiles_n, tn_ = tile - tm_ * tiles_n;                                      \
    const unsigned row_b_ = (unsigned)p.cout_store * 4u;                                             \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                                 \
      unsigned off_ = SP_BASE(tm_, tn_, j);                                                          \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
          rr[i][j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off_, 0, 0));   \
          off_ += ((r & 3) == 3 ? 5u : 1u) * row_b_;      /* 8 (r / 4) + r % 4: three single steps, then one of five */ \
        }                                                                                            \
    }                                                                                                \
  }
  unsigned out_bits = 0;        // bits of the largest output magnitude of this lane so far (the next layer's activation scale)
  int tile = slot;
  for (;;) {
    const int ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);
    for (int c = 0; c < n_chunks; ++c) {
      // tile hand-over of the fetch streams: they enter the next tile three chunks before the MFMAs do (the queue slot
      // was written during chunk 1 and published by the synchronisations since); HALO: the patch stream enters it with
      // the tile's last slice, nine chunks ahead
      if (c == n_chunks - (HALO ? 9 : 3)) {
        int nv;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nv) : "v"(slot_addr) : "memory");
        next_tile = __builtin_amdgcn_readfirstlane(nv);
        if constexpr (HALO) {
          SP_H_SETUP(next_tile);
          f_slice = 0;
        } else {
          fa_tile = next_tile;
          SP_A_SETUP();
        }
      }
      if (c == n_chunks - W_AHEAD) {
        fw_tile = next_tile;
        SP_W_SETUP();
      }
      if constexpr (!HALO) SP_A_TAP();
      if constexpr (EARLY_RES) {
        if (c == n_chunks - 1) SP_RES_REQUEST();
      }
      // first k-step: set 0; its slot 0 reads the second k-step of the same chunk into set 1
      SP_READ_X(1, rd_a);
      SP_STEP(0, 1, true)
      // second k-step: set 1; reads the first k-step of the next chunk (published by this chunk's barrier) into set 0
      rd_a = rd_a + A_STAGE == A_RING * A_STAGE ? 0u : rd_a + A_STAGE;
      rd_w = rd_w + W_STAGE == W_RING * W_STAGE ? 0u : rd_w + W_STAGE;
      if constexpr (HALO) {     // the read side moves to the next chunk: next tap, or the next slice's patch buffer
        if (++r_tap == 9) {
          r_tap = 0;
          r_buf ^= (unsigned)A_STAGE;
          if (c == n_chunks - 1) SP_H_MASK(next_tile);
        }
      }
      landed_target += 8u;
      SP_AWAIT(landed_seen, landed_addr, landed_target);     // every wave's pieces of the next chunk are in LDS
      if constexpr (HALO) {
        // every piece of the patch being fetched (issued with taps 0..4) has landed for every wave two chunks ago: split it;
        // the LANDED signal of the next chunk publishes the pieces (written by now) before the slice's first read
        if (c_tap == 7) {
          SP_H_CONVERT(h_buf);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      SP_READ_X(0, rd_a);
      SP_STEP(1, 0, false)
      if constexpr (HALO) {
        prev_had_patch = c_tap < NPH;
        if (++c_tap == 9) {     // the MFMAs move to the next slice: the buffer they leave takes the patch after next
          c_tap = 0;
          h_buf ^= (unsigned)A_STAGE;
          ++c_slice;
          if (c_slice == n_slices) c_slice = 0;
          f_slice = c_slice + 1 < n_slices ? c_slice + 1 : 0;
        }
      } else {
        SP_A_ADVANCE();
      }
      SP_W_ADVANCE();
    }
    // ---- epilogue: 1 / (weight scale) x accumulator + bias + residual, ReLU, store.  Every residual request goes out
    // before the first store: a load issued behind a store would make the compiler wait for that store (one counter, loads
    // and stores complete out of order with each other) - a write round trip per accumulator.
    {
      const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
      const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();
      float bb[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) bb[j] = p.bias[tn * BN + wn * (NI * 32) + j * 32 + fr];
      if constexpr (!EARLY_RES) SP_RES_REQUEST();
      // skip_waits below relies on every transfer issued so far having landed once the epilogue has its operands.  Without
      // EARLY_RES the residual requests are the youngest operations, so the waits for them say so; with EARLY_RES they went
      // out in front of the last chunk's transfers: wait for everything here (bias and residual are needed now anyway).
      if constexpr (EARLY_RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned row_b = (unsigned)p.cout_store * 4u;
      const bool ragged = tm * BM + BM > M;        // wave-uniform: only then the maximum needs the per-pixel mask
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        unsigned off = SP_BASE(tm, tn, j);
        const unsigned keep_n = off != HOOB ? 0x7FFFFFFFu : 0u;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned o = __float_as_uint(fmaxf(fmaf(acc[i][j][r], tot_unscale, bb[j] + rr[i][j][r]), floor_v));
            const unsigned keep = ragged ? (tm * BM + SP_PIX(i, r) < M ? keep_n : 0u) : keep_n;     // pixels / channels beyond the tensor do not count
            out_bits = max(out_bits, o & keep);
            __builtin_amdgcn_raw_buffer_store_b32(o, o_rsrc, off, 0, 0);
            off += ((r & 3) == 3 ? 5u : 1u) * row_b;
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
      }
      skip_waits = 2;
    }
    if ((unsigned)next_tile >= (unsigned)n_tiles) break;     // (unsigned: a corrupt queue word cannot keep the loop alive)
    tile = next_tile;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the transfers issued for tiles that do not exist
  if (p.out_max) publish_abs_max(p.out_max, out_bits);
#undef SP_H_SETUP
#undef SP_H_ISSUE
#undef SP_H_MASK
#undef SP_H_CONVERT
#undef SP_A_SETUP
#undef SP_W_SETUP
#undef SP_A_TAP
#undef SP_A_ADDR
#undef SP_A_ISSUE
#undef SP_A_ADVANCE
#undef SP_W_ISSUE
#undef SP_W_ADVANCE
#undef SP_READ_X
#undef SP_READ_WALL
#undef SP_READ_W1
#undef SP_CONV
#undef SP_PIN
#undef SP_MFMA
#undef SP_SLOT
#undef SP_RES_REQUEST
#undef SP_PIX
#undef SP_BASE
#undef SP_SIGNAL
#undef SP_AWAIT
#undef SP_PEEK
#undef SP_SLOTS3
#undef SP_STEP
}

template <int BM, int BN, int WR, int WC, bool HALO>
hipError_t launch_split_cfg(const ConvLaunch& c, hipStream_t s) {
  const int M = c.n_img * c.Ho * c.Wo;
  const int tiles_m = (M + BM - 1) / BM;
  const int tiles_n = (c.cout_store + BN - 1) / BN;
  const int n_tiles = tiles_m * tiles_n;
  const size_t lds = (HALO ? 2 * (size_t)320 * 128 : 3 * (size_t)BM * 128) + (HALO ? 4 : 3) * (size_t)BN * 128 + 256 + 16;
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<BM, BN, WR, WC, HALO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL((conv_split_kernel<BM, BN, WR, WC, HALO>), dim3(grid), dim3(512), lds, s, c, tiles_n, n_tiles);
  return hipGetLastError();
}

}  // namespace

bool conv_split_applicable(const ConvLaunch& c) {
  return c.w_split && c.split_unscale > 0.f && c.cslice == 32 && c.cin % 32 == 0 && !c.out_nchw && c.splits == 0 && c.k_pad / 32 >= 6 &&
         c.cout_store % 4 == 0 && c.cout_store >= 64 && c.tile_counter && c.num_cu > 0;
}

hipError_t launch_conv_split(const ConvLaunch& c, hipStream_t s) {
  if (!conv_split_applicable(c)) return hipErrorInvalidValue;
  if ((size_t)c.n_img * c.H * c.W * c.cin * sizeof(float) >= 0x7FFFFF00ull) return hipErrorInvalidValue;
  if ((size_t)c.n_img * c.Ho * c.Wo * c.cout_store * sizeof(float) >= 0x7FFFFF00ull) return hipErrorInvalidValue;
  // the stride-1 convolutions of layer2 .. layer4 and the stride-2 entries of layer3 / layer4 on whole-map tiles: four waves, weights
  // global -> registers, the patch at padded image coordinates, no chunk synchronisation (conv_w4.hip)
  if (conv_w4_applicable(c)) return launch_conv_w4(c, s);
  // layer2's 64 -> 64 convolutions with the weights resident in registers (conv_c64k.hip): the form conv_w4 replaced, kept for A/B
  if (conv_c64k_applicable(c)) return launch_conv_c64k(c, s);
  // stride-1 3x3 from 64 input channels up (image rows of at most 31 pixels): the input halo resident in LDS
  const bool halo = c.ksize == 3 && c.stride == 1 && c.pad == 1 && c.cin >= 64 && c.W <= 31 && c.H == c.Ho && c.W == c.Wo;
  if (c.cout_store <= 64) return halo ? launch_split_cfg<256, 64, 8, 1, true>(c, s) : launch_split_cfg<256, 64, 8, 1, false>(c, s);
  return halo ? launch_split_cfg<256, 128, 4, 2, true>(c, s) : launch_split_cfg<256, 128, 4, 2, false>(c, s);
}

// Host side: the two fp16 planes of the packed fp32 weights [cout_pad][k_pad], pre-scaled by `scale` (a power of two),
// in fragment order: [cout_pad / 32][k_pad / 32][k-step 2][plane 2][lane 64][8] fp16, lane = (row & 31) + 32 * half,
// element e of the lane = k 16 * step + 8 * half + e of the chunk.  Returns the number of 16-bit words (2 * cout_pad * k_pad).
size_t pack_split_weights(const float* w, int cout_pad, int k_pad, float scale, uint16_t* out) {
  const int n_chunks = k_pad / 32;
  for (int grp = 0; grp < cout_pad / 32; ++grp)
    for (int kc = 0; kc < n_chunks; ++kc)
      for (int st = 0; st < 2; ++st)
        for (int ln = 0; ln < 64; ++ln)
          for (int e = 0; e < 8; ++e) {
            const float x = w[(size_t)(grp * 32 + (ln & 31)) * k_pad + kc * 32 + 16 * st + 8 * (ln >> 5) + e] * scale;
            const _Float16 h0 = (_Float16)x;                  // round to nearest even
            const _Float16 h1 = (_Float16)(x - (float)h0);    // exact difference, rounded once
            const size_t base = ((((size_t)grp * n_chunks + kc) * 2 + st) * 2) * 512 + (size_t)ln * 8 + e;
            __builtin_memcpy(&out[base], &h0, 2);
            __builtin_memcpy(&out[base + 512], &h1, 2);
          }
  return (size_t)2 * cout_pad * k_pad;
}

// the power of two that puts the largest weight magnitude in [2^14, 2^15)
float split_weight_scale(const float* w, size_t n) {
  float m = 0.f;
  for (size_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(w[i]));
  if (!(m > 0.f) || !(m < 3.0e38f)) return 1.f;
  int e;
  frexpf(m, &e);                      // m = f * 2^e, f in [0.5, 1)
  return ldexpf(1.f, 15 - e);
}

}  // namespace ut
