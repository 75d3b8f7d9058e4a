// MEASURED OUT (round 2) - kept under tools/diag as the record of the experiment, not built into the library.
// On MI355X this structure reached 115-119 TFLOP/s on the 128->128 @12x12 and 256->256 @6x6 layers (4096 crops) against
// 135-138 for conv_igemm.hip in the same interleaved A/B run (tools/diag/conv_ab.py ... -DDIAG_WS), 99 against 122 on
// 64->64 @24x24; with the loaders issuing nothing (-DWS_NO_DMA, timing only) the four lone MFMA waves still stop at
// 125-128: one MFMA wave per SIMD loses more at the per-chunk barrier than a second co-resident wave costs in issue
// slots.  Results were bit-identical to conv_igemm's.
//
// Wave-specialised implicit-GEMM convolution for gfx950 (exact-fp32 matrix cores, v_mfma_f32_32x32x2_f32):
// two LOADER waves + four MFMA waves per workgroup, one workgroup per CU.
//
// Same math, operand layout and K order as conv_igemm.hip (see its header and ut_kernels.h): NHWC activations,
// weights [cout_pad][k_pad] with BatchNorm folded, K walked in chunks of 32 = one tap of one 32-channel slice,
// operands staged global -> LDS by buffer_load ... lds into XOR-swizzled 128-byte rows, accumulators start at
// bias (+ residual), epilogue = (ReLU) + 16-byte NHWC stores.  It covers the backbone's 3x3 convolutions whose
// channel slice equals the chunk width (cin % 32 == 0) and whose K has at least NS chunks.
//
// What is different: in conv_igemm every wave both feeds the matrix pipe and issues its share of the LDS-DMA pieces
// (address arithmetic + ~60-185 cycles of issue per piece, in order, in front of its own next MFMA), and two or
// three workgroups per CU cover each other's gaps.  Here the four MFMA waves (one per SIMD, 64x64 outputs each) run
// nothing but fragment reads, MFMAs and one barrier per chunk; a fifth wave computes every gather address, issues
// all (BM+BN)/8 pieces of a chunk and keeps a ring of NS = 3 stages full, two chunks ahead of the consumers.
//
// Hand-shake (one s_barrier per chunk, B_g, placed in front of the consumers' last MFMA group of chunk g):
//   consumer: ... MFMA groups 0-2 of chunk g, all fragment reads of chunk g done -> B_g -> first fragment read of
//             chunk g+1, MFMA group 3 of chunk g
//   loader:   (pieces up to chunk g+2 issued) wait until chunk g+1 has landed: s_waitcnt vmcnt(P), the P pieces of
//             chunk g+2 stay in flight -> B_g -> issue chunk g+3 into the stage chunk g just vacated
// The loader also owns the tile queue (first round: static XCD-contiguous slots, then one atomic ticket per tile,
// requested a tile ahead) and publishes the workgroup's tile sequence through an LDS ring.
#include <atomic>

#include "ut_kernels.h"

namespace ut {
bool conv_ws_applicable(const ConvLaunch& c);
hipError_t launch_conv_ws(const ConvLaunch& c, hipStream_t s);
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;

constexpr int BK = 32;
constexpr int LDS_ROW = BK;      // floats per LDS row (128 B), 16-byte slots XOR-swizzled by (row >> 1) & 7
constexpr int NS = 3;            // stages of the operand ring
constexpr unsigned OOB = 0xFFFFFF00u;

// One LDS-DMA piece: 64 lanes x 16 bytes from a buffer (per-lane byte offset, out-of-range -> zeros) straight into
// LDS at lds_addr + lane*16 (see conv_igemm.hip::dma16 for why this is inline asm).
__device__ __forceinline__ void ws_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc)
      : "memory");
}
#ifdef WS_NO_DMA      /* tools/diag timing ablation: the loaders issue nothing (results are wrong) */
#define WS_DMA(r, a, o) asm volatile("" ::"v"(o))
#else
#define WS_DMA(r, a, o) ws_dma16(r, a, o)
#endif
__device__ __forceinline__ u32x4 ws_rsrc_words(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ int ws_fast_div(int n, int d, float inv_d) {   // exact for 0 <= n < 2^24
  int q = (int)((float)n * inv_d);
  int r = n - q * d;
  if (r < 0) --q;
  if (r >= d) ++q;
  return q;
}

template <int WR, int WC>
__global__ __launch_bounds__(384, 2) void conv_ws_kernel(ConvLaunch p, int tiles_n, int n_tiles) {
  static_assert(WR * WC == 4, "four MFMA waves");
  constexpr int BM = 64 * WR, BN = 64 * WC;
  constexpr int MI = 2, NI = 2;                 // 32x32 accumulator tiles per MFMA wave: 64 x 64 outputs
  constexpr int NL = 2;                         // loader waves: loader L issues the pieces of parity L
  constexpr int PA = BM / 8 / NL, PB = BN / 8 / NL;   // 1-KB pieces (8 rows of 128 B) per chunk and loader wave
  constexpr int P = PA + PB;
  constexpr int STAGE = (BM + BN) * LDS_ROW;    // floats
  static_assert(P < 64, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int M = p.n_img * p.Ho * p.Wo;
  const int hw = p.Ho * p.Wo;
  const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)p.Wo;
  const int taps = p.ksize * p.ksize;
  const int n_chunks = p.k_pad / BK;            // >= NS (host check)
  const unsigned smem_addr = (unsigned)(unsigned long)(lds_f32*)smem;
  const unsigned seq_addr = smem_addr + (unsigned)(NS * STAGE * 4);     // int[4]: the workgroup's tile sequence

  // XCD-aware first round: workgroups b, b+8, ... share an XCD (and its L2); give each XCD a contiguous run of tiles
  const int grid = gridDim.x;
  int slot = blockIdx.x;
  if ((grid & 7) == 0) slot = (blockIdx.x & 7) * (grid >> 3) + (blockIdx.x >> 3);

  if (wave >= 4) {
    // =========================================================================================== loader waves
    const int L = wave - 4;                          // this loader's piece parity
    const int sub = lane >> 3, part = lane & 7;      // row inside a piece, 16-byte slot inside the row
    const u32x4 a_words = ws_rsrc_words(p.in, (unsigned)((size_t)p.n_img * p.H * p.W * p.cin * sizeof(float)));
    const u32x4 b_words = ws_rsrc_words(p.w, (unsigned)((size_t)p.cout_pad * p.k_pad * sizeof(float)));
    const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.tile_counter, 0, 4, 0x00020000);
    const unsigned q_off = (lane == 0 && L == 0) ? 0u : OOB;     // only loader 0's lane 0 draws tickets
    int a_pix[PA], a_iy[PA], a_ix[PA];
    unsigned b_off[PB];
    int i_tile = slot, i_chunk = 0, i_tap = 0, i_slice = 0, i_stage = 0, seq_k = 0;
    bool i_valid = true;                             // grid <= n_tiles: the first tile exists
    int ticket;

#define LD_SETUP(TILE)                                                                                \
  {                                                                                                   \
    const int tm_ = (TILE) / tiles_n, tn_ = (TILE) - tm_ * tiles_n;                                   \
    _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                  \
      const int r = 8 * (NL * i + L) + sub;                                                           \
      const int m = tm_ * BM + r;                                                                     \
      const bool ok = m < M;                                                                          \
      const int mm = ok ? m : 0;                                                                      \
      const int img = ws_fast_div(mm, hw, inv_hw);                                                    \
      const int rem = mm - img * hw;                                                                  \
      const int oy = ws_fast_div(rem, p.Wo, inv_wo), ox = rem - oy * p.Wo;                            \
      a_iy[i] = ok ? oy * p.stride - p.pad : -100000;   /* rows beyond M never pass the bounds test */\
      a_ix[i] = ox * p.stride - p.pad;                                                                \
      a_pix[i] = ((img * p.H + a_iy[i]) * p.W + a_ix[i]) * p.cin + 4 * (part ^ ((r >> 1) & 7));       \
    }                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < PB; ++i) {                                                  \
      const int r = 8 * (NL * i + L) + sub;                                                           \
      b_off[i] = (unsigned)(((tn_ * BN + r) * p.k_pad + 4 * (part ^ ((r >> 1) & 7))) * 4);            \
    }                                                                                                 \
    i_chunk = 0; i_tap = 0; i_slice = 0;                                                              \
  }
    // all P pieces of the chunk at the issue pointer -> stage i_stage; then advance the pointer (and the tile)
#define LD_ISSUE()                                                                                    \
  {                                                                                                   \
    int dy = 0, dx = 0;                                                                               \
    if (p.ksize == 3) { dy = (i_tap * 11) >> 5; dx = i_tap - 3 * dy; }                                \
    const int tap_off = (dy * p.W + dx) * p.cin + i_slice * BK;                                       \
    const unsigned dst = smem_addr + (unsigned)(i_stage * STAGE * 4);                                 \
    _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                  \
      const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;                                                 \
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;                   \
      WS_DMA(a_words, dst + 8 * (NL * i + L) * LDS_ROW * 4, ok ? (unsigned)(a_pix[i] + tap_off) * 4u : OOB); \
    }                                                                                                 \
    const unsigned kb = (unsigned)(i_chunk * BK * 4);                                                 \
    _Pragma("unroll") for (int i = 0; i < PB; ++i)                                                    \
      WS_DMA(b_words, dst + (BM + 8 * (NL * i + L)) * LDS_ROW * 4, b_off[i] + kb);                    \
    i_stage = i_stage == NS - 1 ? 0 : i_stage + 1;                                                    \
    ++i_chunk;                                                                                        \
    if (++i_tap == taps) { i_tap = 0; ++i_slice; }                                                    \
    if (i_chunk == 2 && L == 0) {       /* loader 0 publishes the tile after this one (the ticket drawn a tile \
                                           ago) and draws the next ticket; read back >= 1 barrier later */ \
      const int nt = __builtin_amdgcn_readfirstlane(grid + ticket);                                   \
      asm volatile("ds_write_b32 %0, %1" ::"v"(seq_addr + 4u * (unsigned)((seq_k + 1) & 3)), "v"(nt) : "memory"); \
      ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(nt < n_tiles ? 1 : 0, q_rsrc, q_off, 0, 0); \
    }                                                                                                 \
    if (i_chunk == n_chunks) {          /* next tile */                                               \
      ++seq_k;                                                                                        \
      int nv;                                                                                         \
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nv) : "v"(seq_addr + 4u * (unsigned)(seq_k & 3)) : "memory"); \
      i_tile = __builtin_amdgcn_readfirstlane(nv);                                                    \
      i_valid = i_tile < n_tiles;                                                                     \
      if (i_valid) LD_SETUP(i_tile);                                                                  \
    }                                                                                                 \
  }

    if (L == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(seq_addr), "v"(i_tile) : "memory");
    LD_SETUP(i_tile);
    ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, q_rsrc, q_off, 0, 0);
    LD_ISSUE();                                     // chunks 0 and 1 (n_chunks >= NS: same tile)
    LD_ISSUE();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P) : "memory");   // chunk 0 landed (chunk 1 in flight)
    __builtin_amdgcn_s_barrier();                   // B_-1
    int issued = 2, passed = -1;                    // chunks issued, index of the last barrier passed
    for (;;) {
      if (!i_valid && passed + 1 >= issued) break;  // every chunk has had its barrier
      // B_passed is behind us: the stage of chunk `passed` is free -> chunk passed + NS
      if (i_valid) {
        LD_ISSUE();
        ++issued;
        // chunk passed+2 must have landed before B_(passed+1); the chunk just issued stays in flight
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      ++passed;
    }
#undef LD_SETUP
#undef LD_ISSUE
    return;
  }

  // ============================================================================================= MFMA waves
  const int wm = wave / WC, wn = wave % WC;
  const int fr = lane & 31;     // fragment row (A/B) == accumulator column (pixel)
  const int fh = lane >> 5;     // k half (A/B) == accumulator row offset 4*fh

  // with no residual the descriptor is empty and every load returns 0
  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.bias), 0, p.res ? (int)((size_t)M * p.cout_store * sizeof(float)) : 0,
      0x00020000);
  const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.bias), 0, (int)((size_t)p.cout_pad * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)((size_t)M * p.cout_store * sizeof(float)), 0x00020000);

  u32x4 res_raw[MI][NI][4];
  u32x4 bias_raw[NI][4];
  f32x16 acc[MI][NI];
  float4 afX[MI], bfX[NI], afY[MI], bfY[NI];

  // bias + residual requests of tile TILE for accumulator (i, j) = (Q / NI, Q % NI).  Safe for TILE >= n_tiles:
  // rows are beyond M, so every offset is out of range and the loads return zeros.
#define C_INIT_LOAD_PART(TILE, Q)                                                                     \
  {                                                                                                   \
    constexpr int i = (Q) / NI, j = (Q) % NI;                                                         \
    const int tm_ = (TILE) / tiles_n, tn_ = (TILE) - tm_ * tiles_n;                                   \
    if constexpr (i == 0) {                                                                           \
      _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4)                                                \
        bias_raw[j][g4] = __builtin_amdgcn_raw_buffer_load_b128(                                      \
            bias_rsrc, (unsigned)(tn_ * BN + wn * (NI * 32) + j * 32 + 8 * g4 + 4 * fh) * 4u, 0, 0);  \
    }                                                                                                 \
    const int m = tm_ * BM + wm * (MI * 32) + i * 32 + fr;                                            \
    const bool m_ok = m < M;                                                                          \
    _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                                \
      const int n = tn_ * BN + wn * (NI * 32) + j * 32 + 8 * g4 + 4 * fh;                             \
      const unsigned off = (m_ok && n < p.cout_store) ? (unsigned)(m * p.cout_store + n) * 4u : OOB;  \
      res_raw[i][j][g4] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0);                   \
    }                                                                                                 \
  }
#define C_INIT_COMBINE()                                                                              \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                      \
    _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                    \
      _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                              \
        acc[i][j][4 * g4 + 0] = __uint_as_float(bias_raw[j][g4].x) + __uint_as_float(res_raw[i][j][g4].x); \
        acc[i][j][4 * g4 + 1] = __uint_as_float(bias_raw[j][g4].y) + __uint_as_float(res_raw[i][j][g4].y); \
        acc[i][j][4 * g4 + 2] = __uint_as_float(bias_raw[j][g4].z) + __uint_as_float(res_raw[i][j][g4].z); \
        acc[i][j][4 * g4 + 3] = __uint_as_float(bias_raw[j][g4].w) + __uint_as_float(res_raw[i][j][g4].w); \
      }
  // fragment reads of k-group q (8 k) of the stage at float offset ST into register set X or Y
#define C_READ(SET, ST, q)                                                                            \
  {                                                                                                   \
    const int koff_ = 4 * ((2 * (q) + fh) ^ ((fr >> 1) & 7));                                         \
    const float* as = smem + (ST) + (wm * (MI * 32) + fr) * LDS_ROW + koff_;                          \
    const float* bs = smem + (ST) + BM * LDS_ROW + (wn * (NI * 32) + fr) * LDS_ROW + koff_;           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) af##SET[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_ROW); \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) bf##SET[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_ROW); \
  }
  // weights are the MFMA "A" operand, pixels the "B" operand: a lane owns ONE pixel and 4 consecutive output
  // channels per register quad (16-byte NHWC accesses); accumulators visited round-robin
#define C_MFMA_STEP(SET, C)                                                                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                    \
      _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf##SET[j].C, af##SET[i].C, acc[i][j], 0, 0, 0);
#define C_MFMA(SET) { C_MFMA_STEP(SET, x) C_MFMA_STEP(SET, y) C_MFMA_STEP(SET, z) C_MFMA_STEP(SET, w) }
#define C_PIN() __builtin_amdgcn_sched_barrier(0)
  // one 16-byte quad (G4) of accumulator (I, J) of tile (e_tm, e_tn): (ReLU) + store
#define C_EPI_PART(I, J, G4)                                                                          \
  {                                                                                                   \
    const int m = e_tm * BM + wm * (MI * 32) + (I) * 32 + fr;                                         \
    const int n = e_tn * BN + wn * (NI * 32) + (J) * 32 + 8 * (G4) + 4 * fh;                          \
    const unsigned off = (m < M && n < p.cout_store) ? (unsigned)(m * p.cout_store + n) * 4u : OOB;   \
    u32x4 pk;                                                                                         \
    pk.x = __float_as_uint(fmaxf(acc[I][J][4 * (G4) + 0], e_floor));                                  \
    pk.y = __float_as_uint(fmaxf(acc[I][J][4 * (G4) + 1], e_floor));                                  \
    pk.z = __float_as_uint(fmaxf(acc[I][J][4 * (G4) + 2], e_floor));                                  \
    pk.w = __float_as_uint(fmaxf(acc[I][J][4 * (G4) + 3], e_floor));                                  \
    __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, off, 0, 0);                                    \
  }
  // last MFMA group of a tile, one accumulator after the other: the quads of the accumulator that has just been
  // finished are stored in the MFMA gaps of the next one
#define C_Q_MFMA(SET, Q, C) acc[(Q) / NI][(Q) % NI] = __builtin_amdgcn_mfma_f32_32x32x2f32(         \
      bf##SET[(Q) % NI].C, af##SET[(Q) / NI].C, acc[(Q) / NI][(Q) % NI], 0, 0, 0)
#define C_Q_STORE(Q, G4) if constexpr ((Q) >= 0) { C_EPI_PART(((Q) < 0 ? 0 : (Q)) / NI, ((Q) < 0 ? 0 : (Q)) % NI, G4); }
#define C_TAIL_Q(SET, Q)                                                                              \
  {                                                                                                   \
    C_Q_MFMA(SET, Q, x); C_PIN(); C_Q_STORE((Q) - 1, 0); C_PIN();                                     \
    C_Q_MFMA(SET, Q, y); C_PIN(); C_Q_STORE((Q) - 1, 1); C_PIN();                                     \
    C_Q_MFMA(SET, Q, z); C_PIN(); C_Q_STORE((Q) - 1, 2); C_PIN();                                     \
    C_Q_MFMA(SET, Q, w); C_PIN(); C_Q_STORE((Q) - 1, 3); C_PIN();                                     \
  }
#define C_TAIL_EPI(SET)                                                                               \
  {                                                                                                   \
    C_TAIL_Q(SET, 0) C_TAIL_Q(SET, 1) C_TAIL_Q(SET, 2) C_TAIL_Q(SET, 3)                               \
    C_Q_STORE(3, 0); C_Q_STORE(3, 1); C_Q_STORE(3, 2); C_Q_STORE(3, 3);                               \
  }
  // every fragment read of the chunk has returned (so its stage may be refilled) -> B_g
#define C_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  // one chunk: on entry set X holds its q = 0 fragments (read behind the previous chunk's barrier)
#define C_CHUNK(ST, STN)                                                                              \
  {                                                                                                   \
    C_READ(Y, ST, 1); C_PIN(); C_MFMA(X); C_PIN();                                                    \
    C_READ(X, ST, 2); C_PIN(); C_MFMA(Y); C_PIN();                                                    \
    C_READ(Y, ST, 3); C_PIN(); C_MFMA(X); C_PIN();                                                    \
    C_BARRIER();                                                                                      \
    C_READ(X, STN, 0); C_PIN(); C_MFMA(Y); C_PIN();                                                   \
  }
#define C_STEP_INIT(SET, C, Q, TILE) { C_INIT_LOAD_PART(TILE, Q); C_PIN(); C_MFMA_STEP(SET, C) C_PIN(); }
  // last chunk of a tile: the next tile's bias / residual requests ride on the steps of the third group, the
  // finished accumulators are stored in the gaps of the fourth
#define C_CHUNK_LAST(ST, STN, TILE)                                                                   \
  {                                                                                                   \
    C_READ(Y, ST, 1); C_PIN(); C_MFMA(X); C_PIN();                                                    \
    C_READ(X, ST, 2); C_PIN(); C_MFMA(Y); C_PIN();                                                    \
    C_READ(Y, ST, 3); C_PIN();                                                                        \
    C_STEP_INIT(X, x, 0, TILE) C_STEP_INIT(X, y, 1, TILE) C_STEP_INIT(X, z, 2, TILE) C_STEP_INIT(X, w, 3, TILE) \
    C_BARRIER();                                                                                      \
    C_READ(X, STN, 0); C_PIN(); C_TAIL_EPI(Y); C_PIN();                                               \
  }

  int tile = slot;
  C_INIT_LOAD_PART(tile, 0) C_INIT_LOAD_PART(tile, 1) C_INIT_LOAD_PART(tile, 2) C_INIT_LOAD_PART(tile, 3)
  __builtin_amdgcn_s_barrier();                     // B_-1: chunk 0 has landed
  int st = 0;                                       // float offset of the current stage
  int seq_k = 0;
  C_READ(X, 0, 0);
  const float e_floor = p.relu ? 0.f : -__builtin_huge_valf();   // 0 with ReLU, -inf without: one v_max, no branch
  for (;;) {
    C_INIT_COMBINE();
    for (int c = 0; c + 1 < n_chunks; ++c) {
      const int stn = st == (NS - 1) * STAGE ? 0 : st + STAGE;
      C_CHUNK(st, stn);
      st = stn;
    }
    // the loader published the next tile before the barrier two chunks back
    ++seq_k;
    int next_v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(next_v) : "v"(seq_addr + 4u * (unsigned)(seq_k & 3)) : "memory");
    const int next = __builtin_amdgcn_readfirstlane(next_v);
    const int e_tm = tile / tiles_n, e_tn = tile - e_tm * tiles_n;
    {
      const int stn = st == (NS - 1) * STAGE ? 0 : st + STAGE;
      C_CHUNK_LAST(st, stn, next);
      st = stn;
    }
    if (next >= n_tiles) break;
    tile = next;
  }
#undef C_INIT_LOAD_PART
#undef C_INIT_COMBINE
#undef C_READ
#undef C_MFMA_STEP
#undef C_MFMA
#undef C_PIN
#undef C_EPI_PART
#undef C_Q_MFMA
#undef C_Q_STORE
#undef C_TAIL_Q
#undef C_TAIL_EPI
#undef C_BARRIER
#undef C_CHUNK
#undef C_STEP_INIT
#undef C_CHUNK_LAST
}

template <int WR, int WC>
hipError_t launch_ws(const ConvLaunch& c, hipStream_t s) {
  constexpr int BM = 64 * WR, BN = 64 * WC;
  const int M = c.n_img * c.Ho * c.Wo;
  const int tiles_m = (M + BM - 1) / BM;
  const int tiles_n = (c.cout_store + BN - 1) / BN;
  const int n_tiles = tiles_m * tiles_n;
  const size_t lds = (size_t)NS * (BM + BN) * LDS_ROW * sizeof(float) + 16;     // ring + tile-sequence words
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long dev_bit = (c.device >= 0 && c.device < 64) ? 1ull << c.device : 0ull;
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit) || !dev_bit) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<WR, WC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  int grid = c.num_cu;          // one workgroup (4 MFMA waves + 1 loader) per CU
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL((conv_ws_kernel<WR, WC>), dim3(grid), dim3(384), lds, s, c, tiles_n, n_tiles);
  return hipGetLastError();
}

}  // namespace

bool conv_ws_applicable(const ConvLaunch& c) {
  return c.cslice == BK && c.cin % BK == 0 && c.k_pad == c.k_total && c.k_pad / BK >= NS && !c.out_nchw &&
         c.cout_store % 64 == 0 && c.cout_pad >= c.cout_store && c.tile_counter != nullptr && c.num_cu > 0 &&
         (size_t)c.n_img * c.H * c.W * c.cin * sizeof(float) < 0x7FFFFF00ull &&
         (size_t)c.n_img * c.Ho * c.Wo * c.cout_store * sizeof(float) < 0x7FFFFF00ull;
}

hipError_t launch_conv_ws(const ConvLaunch& c, hipStream_t s) {
  if (!conv_ws_applicable(c)) return hipErrorInvalidValue;
  if (c.cout_store % 128 == 0) return launch_ws<2, 2>(c, s);     // 128 x 128 tiles
  return launch_ws<4, 1>(c, s);                                   // 256 x 64 tiles (cout = 64)
}

}  // namespace ut
