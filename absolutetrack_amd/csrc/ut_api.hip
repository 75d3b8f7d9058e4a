// C ABI of libumetrack_hip.so (include/umetrack_hip.h): handle, weight folding/packing, workspace,
// temporal state and the launch sequences of the hot path.  Host-only logic; kernels live in the
// other translation units.
#include "../../include/umetrack_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "ut_kernels.h"

namespace {

thread_local std::string g_create_error;

struct ConvW {
  float* w = nullptr;     // device [cout_pad][k_pad]
  void* w_split = nullptr;   // device: the two fp16 planes of w * 2^k in fragment order (conv_split.hip), eligible layers only
  float split_unscale = 0.f; // 2^-k
  float wsum_rows = 0.f;     // max over output channels of sum_k |w| (folded), rounded up: bounds |conv(x)| by wsum_rows * max|x|
  float bias_max = 0.f;      // max |bias| (folded)
  float* bias = nullptr;  // device [cout_pad]
  int cin = 0, cin_pad = 0, cout = 0, cout_pad = 0, cout_store = 0;
  int taps = 1, ksize = 1, stride = 1, pad = 0, k_total = 0, k_pad = 0, cslice = 0;
  double flops_per_pixel = 0;   // 2 * taps * cin * cout, un-padded
};

struct Block {
  ConvW conv1, conv2, ds;
  bool has_ds = false;
};

struct Regressor {
  Block blocks[2];
  Block blocks_split[2];    // the same convolutions with input and output channels zero-padded to 128: the shape conv_w4.hip takes
  float* w_out = nullptr;   // [D][C] raw (applied after the average pool)
  float* b_out = nullptr;
  int c = 0, d = 0;
};

struct ProfEvent { hipEvent_t a, b; double flops;   int kind = 0;   // 0: fp32 matrix-instruction kernels, 1: split-fp16 kernels
};

}  // namespace

constexpr int kDefaultChunk = 4096;
constexpr int kMaxChunk = 7281;   // 48*48*32*4 B per crop under 2^31 - 256 bytes

struct ut_context {
  int device = 0;
  int num_cu = 256;
  std::string err;
  std::vector<void*> allocs;        // everything to hipFree at destroy
  // weights
  float* stem_w = nullptr; float* stem_b = nullptr;
  Block bb[12];
  ConvW proj, fus0, fus1, fus2, tmp[3];
  float *skel_w = nullptr, *skel_b = nullptr, *skel_scale = nullptr, *skel_shift = nullptr;
  Regressor reg_k, reg_u;
  // backbone workspace.  Phase A (stem, layer1, layer2) runs in passes of `chunk` crops: bounded by the 32-bit
  // byte offsets of the buffer descriptors (a 48x48x32 fp32 map is 295 KB per crop -> at most 7281 crops) and
  // by memory (4.2 GB of workspace at 4096); the convolutions are matrix-pipe bound, so fewer, larger launches
  // win over cache residency (measured: 1024 -> 4096 crops per pass = +1.6 % end to end).  Phase B (layer3,
  // layer4, projection) runs over up to PHASE_B_MAX crops at once so that the small late maps still fill the
  // chip with workgroups.
  int chunk = kDefaultChunk;
  int ws_crops = 0;       // phase-A capacity (crops)
  float *bufX = nullptr, *bufH = nullptr, *bufY = nullptr, *bufD = nullptr;
  // fused resample -> backbone: the crops between the two kernels (u8 grey levels, or fp32 in UT_REMAP_FLOAT mode)
  size_t crops_ws_bytes = 0;
  void* crops_ws = nullptr;
  int wsb_crops = 0;      // phase-B capacity (crops)
  float *bufL2 = nullptr, *bufP = nullptr, *bufQ = nullptr, *bufBH = nullptr, *bufBD = nullptr;
  // head workspace
  int ws_samples = 0;
  ut::HeadBuffers hb{};
  int ws_skel = 0;
  // temporal state
  int slots_cap = 0, slots_used = 0;
  float *mem = nullptr, *prev_ext = nullptr;
  // per-launch device words, zeroed by one memset at the start of a call: [i] the tile queue of launch i of the call,
  // [kMaxCounters + i] the bits of the largest magnitude that launch stored (the activation scale of a split-fp16 consumer)
  unsigned* counters = nullptr;
  int counter_next = 0;
  // split-fp16 activation scales.  Tensor ids: 0 the stem's output, 1 + 2b the output of block b's first convolution, 2 + 2b
  // block b's output (b = 0 .. 11).  calib[t]: the bits of 2^kCalibHeadroom x the largest magnitude of tensor t over the
  // calibration crops (device words behind the per-launch words; never zeroed by begin_call).
  unsigned* calib = nullptr;
  int scale_mode = UT_SPLIT_SCALE_CALIBRATED;
  bool calibrated = false;
  bool head_calibrated = false;     // ... including the regressor's tensors (needs at least two calibration crops)
  bool calibrating = false;         // the running backbone call is a calibration pass: dynamic scales, maxima merged into calib
  unsigned word_gen = 0;            // bumped by every zeroing of the words: a max word kept across launches is stale after it
  bool block_fusion = true;         // split-fp16 mode: layer1's BasicBlocks as one launch each (ut_set_block_fusion)
  int resident_weights = 1;         // split-fp16 mode (ut_set_resident_weights): 1 conv_w4 wherever it applies, 0 the chunked kernels, 2 .. 6 A/B mixes
  bool call_split = false;          // the running backbone call uses the split-fp16 kernels (decided once per call)
  // index checks: device status words ([0] sticky errors, [1] per call), their pinned host mirror, the duplicate-slot
  // scratch (slots_cap ints, allocated with the temporal state) and the mode (UT_CHECK_*)
  int* status = nullptr;
  int* status_host = nullptr;
  int* slot_seen = nullptr;
  int check_mode = UT_CHECK_SYNC;
  // backbone lanes (ut_set_backbone_lanes): a large batch runs as two half-batches on two internal streams, so that
  // the tail of one lane's launch (workgroups that found the tile queue empty) is filled by the other lane's launch
  int lanes = 1;
  hipStream_t lane_stream[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  // UT_CONV_FP32 (exact fp32 matrix instructions) or UT_CONV_SPLIT_F16 (conv_split.hip on the eligible layers)
  int conv_arith = UT_CONV_FP32;
  // latency mode (ut_set_latency_mode): launches with far fewer tiles than CUs split K across workgroups
  bool latency_mode = false;
  float* splitk_ws = nullptr;      // [splits][M][cout] partial sums
  size_t splitk_floats = 0;
  float* zero_bias = nullptr;      // 256 zeros
  // profiling
  bool profiling = false;
  std::vector<ProfEvent> prof;
};

namespace {

int fail(ut_handle h, int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof buf, "%s", what);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(h, x)                                            \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) return fail(h, UT_E_HIP, #x, e_);     \
  } while (0)

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Every entry that allocates or launches runs on the handle's device whatever the caller's current device is,
// and leaves the caller's current device as it found it.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int dev) {      // dev < 0: stay on the caller's current device
    if (dev < 0) return;
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
#define ON_DEVICE_IF(h)                                        \
  DeviceScope scope_((h) ? (h)->device : -1);                  \
  if ((h) && scope_.err != hipSuccess) return fail(h, UT_E_HIP, "hipSetDevice", scope_.err)
#define ON_DEVICE_OF(h)                 \
  DeviceScope scope_((h)->device);      \
  if (scope_.err != hipSuccess) return fail(h, UT_E_HIP, "hipSetDevice", scope_.err)

// Status words for the stateless entry points (handle == NULL): one pair per device, created on first use.
struct DevStatus { int* dev = nullptr; int* host = nullptr; };
std::mutex g_status_mutex;
DevStatus g_status[64];

int stateless_status(int* device_out, DevStatus* out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(nullptr, UT_E_HIP, "hipGetDevice", e);
  if (dev < 0 || dev >= 64) return fail(nullptr, UT_E_INVALID, "device index beyond 63");
  std::lock_guard<std::mutex> lock(g_status_mutex);
  DevStatus& st = g_status[dev];
  if (!st.dev) {
    void *d = nullptr, *hst = nullptr;
    if ((e = hipMalloc(&d, 2 * sizeof(int))) != hipSuccess) return fail(nullptr, UT_E_HIP, "hipMalloc", e);
    if ((e = hipHostMalloc(&hst, 2 * sizeof(int), hipHostMallocDefault)) != hipSuccess) {
      (void)hipFree(d);
      return fail(nullptr, UT_E_HIP, "hipHostMalloc", e);
    }
    if ((e = hipMemset(d, 0, 2 * sizeof(int))) != hipSuccess) return fail(nullptr, UT_E_HIP, "hipMemset", e);
    st.dev = (int*)d; st.host = (int*)hst;
  }
  *device_out = dev;
  *out = st;
  return UT_OK;
}

const char* status_message(int bits) {
  if (bits & ut::UT_SPLIT_RANGE)
    return "range check: an activation entering a split-fp16 convolution is an infinity or a NaN, or lies beyond the calibrated range "
           "of its layer (32 x the calibration maximum; ut_calibrate_split with representative crops, or UT_SPLIT_SCALE_DYNAMIC)";
  if (bits & ut::UT_BAD_SRC_INDEX) return "index check: src_index outside [0, n_src_images)";
  if (bits & ut::UT_BAD_SAMPLE_RANGE) return "index check: sample_range rows must select 1 or 2 crops inside [0, n_crops]";
  if (bits & ut::UT_BAD_MEMORY_IDX) return "index check: memory_idx outside [0, n_slots)";
  if (bits & ut::UT_DUP_MEMORY_IDX) return "index check: memory_idx names one temporal slot twice";
  if (bits & ut::UT_BAD_HAND_IDX) return "index check: hand_idx must be 0 (left) or 1 (right)";
  return "index check: failed";
}

// Read the status words back (synchronises the stream), clear the sticky word when it holds an error.
int read_status(ut_handle h, int* dev, int* host, hipStream_t s, int* sticky, int* call) {
  HIPCHK(h, hipMemcpyAsync(host, dev, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  *sticky = host[0];
  *call = host[1];
  if (*sticky) HIPCHK(h, hipMemsetAsync(dev, 0, sizeof(int), s));
  return UT_OK;
}

struct Cursor {
  const float* p;
  size_t left;
  const float* take(size_t n) {
    if (n > left) { left = 0; ok = false; return nullptr; }
    const float* r = p; p += n; left -= n; return r;
  }
  bool ok = true;
};

struct BN { const float *g = nullptr, *b = nullptr, *m = nullptr, *v = nullptr; };

BN take_bn(Cursor& c, int ch) {
  BN bn;
  bn.g = c.take(ch); bn.b = c.take(ch); bn.m = c.take(ch); bn.v = c.take(ch);
  c.take(1);   // num_batches_tracked
  return bn;
}

int upload(ut_handle h, const std::vector<float>& host, float** dev) {
  void* d = nullptr;
  HIPCHK(h, hipMalloc(&d, host.size() * sizeof(float)));
  h->allocs.push_back(d);
  HIPCHK(h, hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  *dev = (float*)d;
  return UT_OK;
}

// One convolution with eval-mode BatchNorm (eps 1e-5) folded in, as the reference's tensors lay it out:
//   y = s*(conv(x)+b-mean)+beta,  s = gamma/sqrt(var+eps)   ->   w[o][c][t] * s[o],  bias[o] = (b-mean)*s+beta
// (fold in double, one rounding to fp32).
struct Folded {
  std::vector<float> w;      // [cout][cin][taps]
  std::vector<float> b;      // [cout]
  int cin = 0, cout = 0, taps = 0;
};

Folded fold_conv(const float* w, const float* conv_bias, const BN* bn, int cin, int cout, int taps) {
  Folded f;
  f.cin = cin; f.cout = cout; f.taps = taps;
  f.w.resize((size_t)cout * cin * taps);
  f.b.resize(cout);
  for (int o = 0; o < cout; ++o) {
    double s = 1.0, shift = conv_bias ? (double)conv_bias[o] : 0.0;
    if (bn) {
      s = (double)bn->g[o] / sqrt((double)bn->v[o] + 1e-5);
      shift = (shift - (double)bn->m[o]) * s + (double)bn->b[o];
    }
    f.b[o] = (float)shift;
    for (size_t i = (size_t)o * cin * taps; i < (size_t)(o + 1) * cin * taps; ++i) f.w[i] = (float)((double)w[i] * s);
  }
  return f;
}

// ---- channel canonicalisation (exact: powers of two only) --------------------------------------------------------------
// relu(bn(conv)) commutes with a positive per-channel factor, and a power of two commutes with every fp32 rounding
// (lib/models/backbone_resnet.py:56-72): scaling output channel c of a producer (its folded weight row and bias) by 2^k
// and input channel c of every consumer (its weight column) by 2^-k leaves every later fp32 value bit for bit as it was.
// A checkpoint fixes the scale of an inner channel only up to that freedom (a near-dead BatchNorm channel and the large
// consumer weights that compensate it are the same function as a well-scaled pair), while the split-fp16 arithmetic keeps
// ONE power-of-two scale per activation tensor and ONE per weight tensor: a channel 2^-18 below its tensor's largest has a
// subnormal second piece.  So every channel is brought to a canonical scale at pack time: 2^k_c puts the largest magnitude
// among the channel's producer rows (weights and bias) into [1, 2).  Two networks that differ by per-channel powers of two
// pack to the same tensors, in both arithmetics.
float row_max(const Folded& f, int o) {
  float m = fabsf(f.b[o]);
  const size_t n = (size_t)f.cin * f.taps;
  for (size_t i = 0; i < n; ++i) {
    const float a = fabsf(f.w[(size_t)o * n + i]);
    m = a > m ? a : m;          // (a NaN never raises m: such a row keeps its scale)
  }
  return m;
}
int octave_shift(float m) {       // k with m * 2^k in [1, 2); 0 when the row is all zeros or not finite
  if (!(m > 0.f) || !(m < INFINITY)) return 0;
  return -ilogbf(m);
}
void scale_row(Folded& f, int o, int k) {
  if (!k) return;
  const size_t n = (size_t)f.cin * f.taps;
  for (size_t i = 0; i < n; ++i) f.w[(size_t)o * n + i] = ldexpf(f.w[(size_t)o * n + i], k);
  f.b[o] = ldexpf(f.b[o], k);
}
void scale_col(Folded& f, int c, int k) {
  if (!k) return;
  for (int o = 0; o < f.cout; ++o)
    for (int t = 0; t < f.taps; ++t) {
      float& v = f.w[((size_t)o * f.cin + c) * f.taps + t];
      v = ldexpf(v, k);
    }
}
// Channel c of one activation tensor: `keys` (a subset of its producers) define 2^k_c, every producer's row c is scaled by it
// and every consumer's column c by its inverse.
void canonicalise_channels(const std::vector<Folded*>& keys, const std::vector<Folded*>& producers,
                           const std::vector<Folded*>& consumers) {
  const int ch = keys[0]->cout;
  for (int c = 0; c < ch; ++c) {
    float m = 0.f;
    for (Folded* p : keys) { const float r = row_max(*p, c); m = r > m ? r : m; }
    const int k = octave_shift(m);
    for (Folded* p : producers) scale_row(*p, c, k);
    for (Folded* q : consumers) scale_col(*q, c, -k);
  }
}

// Pack folded weights to [cout_pad][k_pad] with k = slice*(taps*cslice) + tap*cslice + c (see ut_kernels.h).
int pack_conv(ut_handle h, ConvW& cw, const Folded& f, int ksize, int stride, int cout_store) {
  const int cin = f.cin, cout = f.cout;
  cw.cin = cin; cw.cout = cout; cw.ksize = ksize; cw.stride = stride;
  cw.pad = ksize == 3 ? 1 : 0;
  cw.taps = ksize * ksize;
  cw.cin_pad = round_up(cin, 4);
  cw.cout_store = cout_store;
  cw.cout_pad = round_up(cout_store, 128);
  cw.k_total = cw.taps * cw.cin_pad;
  cw.cslice = cw.cin_pad % 32 == 0 ? 32 : cw.cin_pad;
  cw.k_pad = round_up(cw.k_total, 32);
  cw.flops_per_pixel = 2.0 * cw.taps * cin * cout;
  std::vector<float> wp((size_t)cw.cout_pad * cw.k_pad, 0.f), bp(cw.cout_pad, 0.f);
  for (int o = 0; o < cout; ++o) {
    bp[o] = f.b[o];
    for (int c = 0; c < cin; ++c)
      for (int t = 0; t < cw.taps; ++t)
        wp[(size_t)o * cw.k_pad + (c / cw.cslice) * (cw.taps * cw.cslice) + t * cw.cslice + c % cw.cslice] =
            f.w[((size_t)o * cin + c) * cw.taps + t];
  }
  {
    double ws = 0.0, bm = 0.0;
    for (int o = 0; o < cout; ++o) {
      double rs = 0.0;
      for (int k = 0; k < cw.k_pad; ++k) rs += fabs((double)wp[(size_t)o * cw.k_pad + k]);
      ws = rs > ws ? rs : ws;
      bm = fabs((double)bp[o]) > bm ? fabs((double)bp[o]) : bm;
    }
    cw.wsum_rows = (float)(ws * 1.0001);
    cw.bias_max = (float)(bm * 1.0001);
  }
  int rc = upload(h, wp, &cw.w);
  if (rc) return rc;
  // fp16 planes for the split-fp16 kernels: the layers they take (channel slice == chunk width, >= 6 chunks: the 3x3
  // convolutions of the backbone; conv_split.hip from 64 channels out, conv_patch.hip's split instantiation for layer1)
  // (and the 1x1 stride-2 shortcut of a 32-channel input: conv_c32s2.hip computes it beside the block's first convolution)
  if (cw.cslice == 32 && (cw.k_pad / 32 >= 6 || (ksize == 1 && stride == 2 && cw.cin_pad == 32)) && cw.cout_store >= 32 && cw.cout_store % 4 == 0) {
    std::vector<uint16_t> planes((size_t)2 * cw.cout_pad * cw.k_pad);
    const float scale = ut::split_weight_scale(wp.data(), wp.size());
    cw.split_unscale = 1.0f / scale;
    ut::pack_split_weights(wp.data(), cw.cout_pad, cw.k_pad, scale, planes.data());
    void* d = nullptr;
    HIPCHK(h, hipMalloc(&d, planes.size() * sizeof(uint16_t)));
    h->allocs.push_back(d);
    HIPCHK(h, hipMemcpy(d, planes.data(), planes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    cw.w_split = d;
  }
  return upload(h, bp, &cw.bias);
}

int pack_conv(ut_handle h, ConvW& cw, const float* w, const float* conv_bias, const BN* bn, int cin, int cout,
              int ksize, int stride, int cout_store) {
  return pack_conv(h, cw, fold_conv(w, conv_bias, bn, cin, cout, ksize * ksize), ksize, stride, cout_store);
}

constexpr int kRegSplitCh = 128;   // channels of the regressor's tensors when its convolutions run in split-fp16 (zero padded)

// the same convolution with zero rows / columns up to cout x cin channels
Folded pad_channels(const Folded& f, int cin, int cout) {
  Folded g;
  g.cin = cin; g.cout = cout; g.taps = f.taps;
  g.w.assign((size_t)cout * cin * f.taps, 0.f);
  g.b.assign(cout, 0.f);
  for (int o = 0; o < f.cout; ++o) {
    g.b[o] = f.b[o];
    for (int c = 0; c < f.cin; ++c)
      for (int t = 0; t < f.taps; ++t) g.w[((size_t)o * cin + c) * f.taps + t] = f.w[((size_t)o * f.cin + c) * f.taps + t];
  }
  return g;
}

// A BasicBlock's folded convolutions.
struct FoldedBlock {
  Folded conv1, conv2, ds;
  bool has_ds = false;
  int stride = 1;
};

bool fold_block(Cursor& c, FoldedBlock& fb, int cin, int cout, int stride, bool ds) {
  const float* w1 = c.take((size_t)cout * cin * 9);
  BN bn1 = take_bn(c, cout);
  const float* w2 = c.take((size_t)cout * cout * 9);
  BN bn2 = take_bn(c, cout);
  const float* wd = nullptr;
  BN bnd;
  if (ds) { wd = c.take((size_t)cout * cin); bnd = take_bn(c, cout); }
  if (!c.ok) return false;
  fb.stride = stride; fb.has_ds = ds;
  fb.conv1 = fold_conv(w1, nullptr, &bn1, cin, cout, 9);
  fb.conv2 = fold_conv(w2, nullptr, &bn2, cout, cout, 9);
  if (ds) fb.ds = fold_conv(wd, nullptr, &bnd, cin, cout, 1);
  return true;
}

// the block's inner channels: conv1 writes them, conv2 reads them (conv1's input channels must have their final scale)
void canonicalise_inner(FoldedBlock& fb) { canonicalise_channels({&fb.conv1}, {&fb.conv1}, {&fb.conv2}); }

int pack_block(ut_handle h, Block& b, const FoldedBlock& fb) {
  int rc;
  const int cs = round_up(fb.conv1.cout, 4);
  if ((rc = pack_conv(h, b.conv1, fb.conv1, 3, fb.stride, cs))) return rc;
  if ((rc = pack_conv(h, b.conv2, fb.conv2, 3, 1, cs))) return rc;
  b.has_ds = fb.has_ds;
  if (fb.has_ds && (rc = pack_conv(h, b.ds, fb.ds, 1, fb.stride, cs))) return rc;
  return UT_OK;
}

// stem + ResNet "2352" + projection, folded, with canonical channel scales
struct FoldedBackbone {
  Folded stem, proj;
  FoldedBlock fb[12];
};

bool fold_backbone(Cursor& c, FoldedBackbone& out) {
  // stem (lib/models/model_utils.py:119-124)
  const float* sw = c.take(32 * 9);
  const float* sb = c.take(32);
  BN sbn = take_bn(c, 32);
  if (!c.ok) return false;
  Folded& stem = out.stem;
  Folded& proj = out.proj;
  FoldedBlock* fb = out.fb;
  stem = fold_conv(sw, sb, &sbn, 1, 32, 9);
  // ResNet layers "2352", planes 32/64/128/256, strides 1/2/2/2 (lib/models/backbone_resnet.py:168-192)
  const int nb[4] = {2, 3, 5, 2}, planes[4] = {32, 64, 128, 256}, strides[4] = {1, 2, 2, 2};
  int first_of_layer[5] = {0, 0, 0, 0, 12};
  int cin = 32, bi = 0;
  for (int l = 0; l < 4; ++l) {
    first_of_layer[l] = bi;
    for (int k = 0; k < nb[l]; ++k) {
      int st = k == 0 ? strides[l] : 1;
      bool ds = k == 0 && (st != 1 || cin != planes[l]);
      if (!fold_block(c, fb[bi++], cin, planes[l], st, ds)) return false;
      cin = planes[l];
    }
  }
  const float* pw = c.take(72 * 256); const float* pb = c.take(72);
  if (!c.ok) return false;
  proj = fold_conv(pw, pb, nullptr, 256, 72, 1);
  // Canonical channel scales, fixed in network order so that each one is defined by tensors whose input side is final
  // already (two checkpoints that differ by per-channel powers of two then arrive at the same tensors):
  //  - the TRUNK of a layer (the tensor its identity shortcuts carry through the blocks): channel c is written by conv2 of
  //    every block of the layer and by the first block's shortcut convolution (layer1: by the stem) and read by conv1 of the
  //    layer's later blocks and by whatever enters the next layer (its first block's conv1 and shortcut convolution; after
  //    layer4: the projection, whose 72 outputs are the features of the ABI and keep their scale).  Its scale comes from the
  //    rows that write the trunk's FIRST tensor: the stem's, or the first block's shortcut and conv2 rows;
  //  - the inner channels of every block, once the block's input has its scale.
  for (int l = 0; l < 4; ++l) {
    const int b0 = first_of_layer[l], b1 = l < 3 ? first_of_layer[l + 1] : 12;
    std::vector<Folded*> keys, prod, cons;
    if (fb[b0].has_ds) {
      canonicalise_inner(fb[b0]);             // its input is the previous layer's trunk: final
      keys = {&fb[b0].ds, &fb[b0].conv2};
      prod = {&fb[b0].ds};
    } else {
      keys = {&stem};
      prod = {&stem};
      cons.push_back(&fb[b0].conv1);
    }
    for (int b = b0; b < b1; ++b) {
      prod.push_back(&fb[b].conv2);
      if (b > b0) cons.push_back(&fb[b].conv1);
    }
    if (l < 3) { cons.push_back(&fb[b1].conv1); cons.push_back(&fb[b1].ds); }
    else cons.push_back(&proj);
    canonicalise_channels(keys, prod, cons);
    for (int b = fb[b0].has_ds ? b0 + 1 : b0; b < b1; ++b) canonicalise_inner(fb[b]);
  }
  return true;
}

int take_block(ut_handle h, Cursor& c, Block& b, int cin, int cout, int stride, bool ds) {
  FoldedBlock fb;
  if (!fold_block(c, fb, cin, cout, stride, ds)) return fail(h, UT_E_WEIGHTS, "weight blob too short");
  canonicalise_inner(fb);
  return pack_block(h, b, fb);
}

int take_regressor(ut_handle h, Cursor& c, Regressor& r, int ch, int d) {
  r.c = ch; r.d = d;
  int rc;
  for (int i = 0; i < 2; ++i) {
    FoldedBlock fb;
    if (!fold_block(c, fb, ch, ch, 1, false)) return fail(h, UT_E_WEIGHTS, "weight blob too short");
    canonicalise_inner(fb);
    if ((rc = pack_block(h, r.blocks[i], fb))) return rc;
    FoldedBlock wide;
    wide.stride = 1; wide.has_ds = false;
    wide.conv1 = pad_channels(fb.conv1, kRegSplitCh, kRegSplitCh);
    wide.conv2 = pad_channels(fb.conv2, kRegSplitCh, kRegSplitCh);
    if ((rc = pack_block(h, r.blocks_split[i], wide))) return rc;
  }
  const float* w = c.take((size_t)d * ch);
  const float* b = c.take(d);
  if (!c.ok) return fail(h, UT_E_WEIGHTS, "weight blob too short");
  if ((rc = upload(h, std::vector<float>(w, w + (size_t)d * ch), &r.w_out))) return rc;
  return upload(h, std::vector<float>(b, b + d), &r.b_out);
}

int dev_alloc(ut_handle h, float** p, size_t n_floats) {
  void* d = nullptr;
  HIPCHK(h, hipMalloc(&d, n_floats * sizeof(float)));
  h->allocs.push_back(d);
  *p = (float*)d;
  return UT_OK;
}

void dev_free(ut_handle h, void* p) {
  if (!p) return;
  for (size_t i = 0; i < h->allocs.size(); ++i)
    if (h->allocs[i] == p) { h->allocs.erase(h->allocs.begin() + i); break; }
  (void)hipFree(p);
}

int ensure_backbone_ws(ut_handle h, int crops) {
  if (crops <= h->ws_crops) return UT_OK;
  HIPCHK(h, hipDeviceSynchronize());
  dev_free(h, h->bufX); dev_free(h, h->bufH); dev_free(h, h->bufY); dev_free(h, h->bufD);
  h->bufX = h->bufH = h->bufY = h->bufD = nullptr;
  h->ws_crops = 0;
  const size_t big = (size_t)crops * 48 * 48 * 32, small = (size_t)crops * 24 * 24 * 64;
  int rc;
  if ((rc = dev_alloc(h, &h->bufX, big)) || (rc = dev_alloc(h, &h->bufH, big)) ||
      (rc = dev_alloc(h, &h->bufY, big)) || (rc = dev_alloc(h, &h->bufD, small)))
    return rc;
  h->ws_crops = crops;
  return UT_OK;
}

constexpr int PHASE_B_MAX = 8192;    // its input, the 24x24x64 map (147 KB per crop), must stay under 2^31 bytes: < 14563 crops

int ensure_phase_b_ws(ut_handle h, int crops) {
  if (crops <= h->wsb_crops) return UT_OK;
  HIPCHK(h, hipDeviceSynchronize());
  float** ptrs[] = {&h->bufL2, &h->bufP, &h->bufQ, &h->bufBH, &h->bufBD};
  for (auto pp : ptrs) { dev_free(h, *pp); *pp = nullptr; }
  h->wsb_crops = 0;
  const size_t l2 = (size_t)crops * 24 * 24 * 64, l3 = (size_t)crops * 12 * 12 * 128;
  int rc;
  if ((rc = dev_alloc(h, &h->bufL2, l2)) || (rc = dev_alloc(h, &h->bufP, l3)) || (rc = dev_alloc(h, &h->bufQ, l3)) ||
      (rc = dev_alloc(h, &h->bufBH, l3)) || (rc = dev_alloc(h, &h->bufBD, l3)))
    return rc;
  h->wsb_crops = crops;
  return UT_OK;
}

int ensure_head_ws(ut_handle h, int samples, int n_skel) {
  int rc;
  if (samples > h->ws_samples) {
    HIPCHK(h, hipDeviceSynchronize());
    float** ptrs[] = {&h->hb.cat144, &h->hb.f108, &h->hb.f72a, &h->hb.f72b, &h->hb.fused, &h->hb.t92a,
                      &h->hb.t92b, &h->hb.regin, &h->hb.rega, &h->hb.regb};
    const int ch[] = {144, 108, 72, 72, 72, 92, 92, kRegSplitCh, kRegSplitCh, kRegSplitCh};
    for (int i = 0; i < 10; ++i) {
      dev_free(h, *ptrs[i]);
      *ptrs[i] = nullptr;
    }
    h->ws_samples = 0;
    for (int i = 0; i < 10; ++i)
      if ((rc = dev_alloc(h, ptrs[i], (size_t)samples * 36 * ch[i]))) return rc;
    h->ws_samples = samples;
  }
  if (n_skel > h->ws_skel) {
    HIPCHK(h, hipDeviceSynchronize());
    dev_free(h, h->hb.skel);
    h->hb.skel = nullptr;
    h->ws_skel = 0;
    if ((rc = dev_alloc(h, &h->hb.skel, (size_t)n_skel * 36 * 4))) return rc;
    h->ws_skel = n_skel;
  }
  return UT_OK;
}

int ensure_slots(ut_handle h, int slots, hipStream_t s) {
  if (slots <= h->slots_cap) return UT_OK;
  int cap = h->slots_cap ? h->slots_cap : 2;
  while (cap < slots) cap *= 2;
  float *nm = nullptr, *ne = nullptr, *seen = nullptr;
  int rc;
  if ((rc = dev_alloc(h, &nm, (size_t)cap * 36 * 18)) || (rc = dev_alloc(h, &ne, (size_t)cap * 16)) ||
      (rc = dev_alloc(h, &seen, (size_t)cap)))
    return rc;
  HIPCHK(h, hipMemsetAsync(nm, 0, (size_t)cap * 36 * 18 * sizeof(float), s));
  HIPCHK(h, hipMemsetAsync(ne, 0, (size_t)cap * 16 * sizeof(float), s));
  if (h->slots_cap) {
    HIPCHK(h, hipMemcpyAsync(nm, h->mem, (size_t)h->slots_cap * 36 * 18 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipMemcpyAsync(ne, h->prev_ext, (size_t)h->slots_cap * 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    dev_free(h, h->mem);
    dev_free(h, h->prev_ext);
    dev_free(h, h->slot_seen);
  }
  h->mem = nm; h->prev_ext = ne; h->slot_seen = (int*)seen; h->slots_cap = cap;
  return UT_OK;
}

constexpr int kMaxCounters = 4096;
constexpr int kScaleTensors = 33;      // 25 of the backbone; 25 .. 28 the known-skeleton regressor's input, inner tensors and first block's output, 29 .. 32 the other regressor's
constexpr int kCalibHeadroom = 4;      // calibrated scale words hold 2^4 x the calibration maximum: inputs up to 32 x that maximum
                                       // (the scale leaves another factor 2 under fp16's 65504) are inside the split's range

// zero the per-launch words (tile queues, output maxima) used by the launches of one API call (stream ordered)
int begin_call(ut_handle h, hipStream_t s) {
  h->counter_next = 0;
  ++h->word_gen;
  HIPCHK(h, ut::launch_zero_words(h->counters, 2 * kMaxCounters, s));      // a kernel, not a memset node: see launch_zero_words
  return UT_OK;
}

// the next launch's index into the per-launch words
int next_launch_word(ut_handle h, hipStream_t s, int* idx) {
  if (h->counter_next >= kMaxCounters) {   // recycle: stream order puts the memset behind the earlier launches
    int rc0 = begin_call(h, s);
    if (rc0) return rc0;
  }
  *idx = h->counter_next++;
  return UT_OK;
}

// The scale word a split-fp16 consumer of tensor `tid` reads, and the word it guards against: calibrated mode - the handle's
// calibrated word, guarded by the word the producer left in this call (may be null); dynamic mode and calibration passes - the
// producer's word itself (null: no scale, the launch stays on the fp32 instruction).
struct ScaleRef { const unsigned* word = nullptr; const unsigned* obs = nullptr; };
ScaleRef scale_for(ut_handle h, int tid, const unsigned* producer_word) {
  ScaleRef r;
  if (!h->call_split || h->latency_mode || tid < 0 || tid >= kScaleTensors) return r;
  if (h->calibrating || h->scale_mode == UT_SPLIT_SCALE_DYNAMIC) r.word = producer_word;
  else { r.word = h->calib + tid; r.obs = producer_word; }
  return r;
}
// calibration pass: fold the producer's word of tensor `tid` into its calibrated word
int note_calibration(ut_handle h, int tid, const unsigned* producer_word, hipStream_t s) {
  if (h->calibrating && producer_word && tid >= 0 && tid < kScaleTensors) HIPCHK(h, ut::launch_merge_max(h->calib + tid, producer_word, s));
  return UT_OK;
}

// in_max: the max word of the launch that produced `in` (null: unknown); in_tid: the tensor id of `in` (< 0: not a tensor of the
// split-fp16 path - the launch stays on the fp32 instruction);
// *out_max (optional): receives this launch's max word when it ran a kernel that leaves one, else null
int run_conv(ut_handle h, const ConvW& cw, const float* in, const float* res, float* out, int n_img, int H, int W,
             bool relu, bool nchw, hipStream_t s, const unsigned* in_max = nullptr, unsigned** out_max = nullptr, int in_tid = -1) {
  if (out_max) *out_max = nullptr;
  ut::ConvLaunch c{};
  c.in = in; c.w = cw.w; c.bias = cw.bias; c.res = res; c.out = out;
  c.n_img = n_img; c.H = H; c.W = W; c.cin = cw.cin_pad;
  c.Ho = (H + 2 * cw.pad - cw.ksize) / cw.stride + 1;
  c.Wo = (W + 2 * cw.pad - cw.ksize) / cw.stride + 1;
  c.cout_store = cw.cout_store; c.cout_pad = cw.cout_pad;
  c.k_total = cw.k_total; c.k_pad = cw.k_pad; c.cslice = cw.cslice;
  c.ksize = cw.ksize; c.stride = cw.stride; c.pad = cw.pad;
  c.relu = relu; c.out_nchw = nchw;
  c.device = h->device; c.num_cu = h->num_cu; { static const int kMask[7] = {15, 0, 14, 9, 12, 13, 8}; c.no_resident = kMask[h->resident_weights]; }      // (ut_kernels.h::ConvLaunch::no_resident)
  int word = 0;
  {
    const unsigned gen = h->word_gen;
    int rc0 = next_launch_word(h, s, &word);
    if (rc0) return rc0;
    if (gen != h->word_gen) in_max = nullptr;      // recycled in mid-call: the producer's word has just been zeroed
  }
  c.tile_counter = h->counters + word;
  // Latency mode: a convolution of a few crops has far fewer 64x64 tiles than the chip has CUs and every workgroup
  // walks all of K alone (a layer-4 conv of 4 crops: 12 tiles x 72 chunks).  Cut K into S equal chunk ranges (S the
  // largest divisor of the chunk count that leaves >= 6 chunks per range and <= one workgroup per CU), let S x tiles
  // workgroups write partial sums and add them in a fixed order afterwards.  Deterministic, but the summation order
  // differs from the unsplit kernel's: results agree with it to fp32 rounding, not bit for bit - hence opt-in.
  int splits = 1;
  if (h->latency_mode && !nchw && cw.cout_store % 4 == 0) {
    const long m = (long)n_img * c.Ho * c.Wo;
    const long tiles64 = ((m + 63) / 64) * ((cw.cout_store + 63) / 64);
    const int chunks = cw.k_pad / 32;
    for (int sp = 2; sp <= chunks / 6; ++sp)
      if (chunks % sp == 0 && tiles64 * sp <= (long)h->num_cu && (size_t)sp * m * cw.cout_store <= h->splitk_floats) splits = sp;
  }
  if (h->latency_mode) c.splits = 1;
  if (splits > 1) {
    ut::ConvLaunch part = c;
    part.bias = h->zero_bias; part.res = nullptr; part.relu = 0; part.out = h->splitk_ws; part.splits = splits;
    HIPCHK(h, ut::launch_conv_igemm(part, s));
    HIPCHK(h, ut::launch_splitk_finish(h->splitk_ws, splits, n_img * c.Ho * c.Wo, cw.cout_store, cw.bias, res, out,
                                       relu ? 1 : 0, s));
    return UT_OK;
  }
  ProfEvent pe{};
  if (h->profiling) {
    HIPCHK(h, hipEventCreateWithFlags(&pe.a, hipEventDisableSystemFence));   // timing only: no system-scope flush per kernel
    HIPCHK(h, hipEventCreateWithFlags(&pe.b, hipEventDisableSystemFence));
    pe.flops = cw.flops_per_pixel * (double)n_img * c.Ho * c.Wo;
    HIPCHK(h, hipEventRecord(pe.a, s));
  }
  // split-fp16 arithmetic: decided once per backbone call (run_backbone), for every eligible layer of the call whose
  // producer left a max word; everything else - the head, and every launch in latency mode - stays on the fp32 instruction
  const ScaleRef sr = scale_for(h, in_tid, in_max);
  c.w_split = sr.word ? cw.w_split : nullptr;
  c.split_unscale = cw.split_unscale;
  c.status = h->status;
  c.in_max = sr.word;
  c.in_obs = sr.obs;
  pe.kind = c.w_split && (ut::conv_split_applicable(c) || ut::conv_patch_applicable(c)) ? 1 : 0;
  if (pe.kind) {
    c.out_max = h->counters + kMaxCounters + word;
    if (out_max) *out_max = c.out_max;
    int rc1 = note_calibration(h, in_tid, in_max, s);
    if (rc1) return rc1;
  }
  if (c.w_split && ut::conv_split_applicable(c)) HIPCHK(h, ut::launch_conv_split(c, s));
  else HIPCHK(h, ut::launch_conv_igemm(c, s));
  if (h->profiling) {
    HIPCHK(h, hipEventRecord(pe.b, s));
    h->prof.push_back(pe);
  }
  return UT_OK;
}

// relu(bn2(conv2(relu(bn1(conv1 x)))) + (downsample(x) | x))   lib/models/backbone_resnet.py:56-72
// x_max: the max word of x's producer (or null); *y_max (optional): the word of the block's output;
// x_tid / mid_tid: the scale-tensor ids of x and of conv1's output (backbone block b: 2 b, 2 b + 1; < 0: no split-fp16 path)
int run_block(ut_handle h, const Block& b, const float* x, float* tmp, float* dsbuf, float* y, int n_img, int H,
              int W, hipStream_t s, const unsigned* x_max = nullptr, unsigned** y_max = nullptr, int x_tid = -1, int mid_tid = -1) {
  int rc;
  if (y_max) *y_max = nullptr;
  // the words of one block (at most three launches) come from one zeroing: a word handed from conv1 to conv2 is never recycled
  // between the two (x's own word is gone after a recycle: its consumers then run unguarded / on the fp32 instruction)
  if (h->counter_next + 8 > kMaxCounters) {
    if ((rc = begin_call(h, s))) return rc;
    x_max = nullptr;
  }
  const ScaleRef xs = scale_for(h, x_tid, x_max);
  // layer1 in split-fp16 mode: the whole block in one launch, the intermediate stays in LDS (conv_block32.hip)
  if (h->block_fusion && xs.word && !b.has_ds && b.conv1.w_split && b.conv2.w_split && b.conv1.stride == 1 &&
      b.conv1.cin_pad == 32 && b.conv1.cout_store == 32 && b.conv2.cout_store == 32) {
    ut::BlockLaunch bl{};
    bl.in = x; bl.out = y; bl.w1_split = b.conv1.w_split; bl.w2_split = b.conv2.w_split;
    bl.unscale_w1 = b.conv1.split_unscale; bl.unscale_w2 = b.conv2.split_unscale;
    bl.bias1 = b.conv1.bias; bl.bias2 = b.conv2.bias;
    bl.wsum1 = b.conv1.wsum_rows; bl.bmax1 = b.conv1.bias_max;
    bl.in_max = xs.word; bl.in_obs = xs.obs; bl.status = h->status;
    bl.n_img = n_img; bl.H = H; bl.W = W; bl.device = h->device; bl.num_cu = h->num_cu;
    if (ut::conv_block32_applicable((bl.tile_counter = h->counters, bl))) {
      const unsigned gen = h->word_gen;
      int word = 0;
      if ((rc = next_launch_word(h, s, &word))) return rc;
      if (gen == h->word_gen) {          // (never a recycle here: the block's words were reserved above)
        if ((rc = note_calibration(h, x_tid, x_max, s))) return rc;
        bl.tile_counter = h->counters + word;
        bl.out_max = h->counters + kMaxCounters + word;
        ProfEvent pe{};
        if (h->profiling) {
          HIPCHK(h, hipEventCreateWithFlags(&pe.a, hipEventDisableSystemFence));
          HIPCHK(h, hipEventCreateWithFlags(&pe.b, hipEventDisableSystemFence));
          pe.flops = (b.conv1.flops_per_pixel + b.conv2.flops_per_pixel) * (double)n_img * H * W;
          pe.kind = 1;
          HIPCHK(h, hipEventRecord(pe.a, s));
        }
        HIPCHK(h, ut::launch_conv_block32(bl, s));
        if (h->profiling) {
          HIPCHK(h, hipEventRecord(pe.b, s));
          h->prof.push_back(pe);
        }
        if (y_max) *y_max = bl.out_max;
        return UT_OK;
      }
    }
  }
  // layer2's entry in split-fp16 mode: the stride-2 3x3 and the 1x1 shortcut from one pass over x (conv_c32s2.hip)
  if (h->block_fusion && xs.word && b.has_ds && b.conv1.w_split && b.ds.w_split && b.conv2.w_split &&
      b.conv1.stride == 2 && b.conv1.cin_pad == 32 && b.conv1.cout_store == 64 && b.ds.cout_store == 64 && dsbuf) {
    ut::Stride2Launch sl{};
    sl.in = x; sl.out1 = tmp; sl.out2 = dsbuf; sl.w1_split = b.conv1.w_split; sl.wd_split = b.ds.w_split;
    sl.unscale1 = b.conv1.split_unscale; sl.unscale_d = b.ds.split_unscale;
    sl.bias1 = b.conv1.bias; sl.bias_d = b.ds.bias;
    sl.in_max = xs.word; sl.in_obs = xs.obs; sl.status = h->status;
    sl.n_img = n_img; sl.H = H; sl.W = W; sl.device = h->device; sl.num_cu = h->num_cu;
    if (ut::conv_c32s2_applicable(sl)) {
      const unsigned gen = h->word_gen;
      int word = 0;
      if ((rc = next_launch_word(h, s, &word))) return rc;
      if (gen == h->word_gen) {
        if ((rc = note_calibration(h, x_tid, x_max, s))) return rc;
        sl.out1_max = h->counters + kMaxCounters + word;
        ProfEvent pe{};
        if (h->profiling) {
          HIPCHK(h, hipEventCreateWithFlags(&pe.a, hipEventDisableSystemFence));
          HIPCHK(h, hipEventCreateWithFlags(&pe.b, hipEventDisableSystemFence));
          pe.flops = (b.conv1.flops_per_pixel + b.ds.flops_per_pixel) * (double)n_img * (H / 2) * (W / 2);
          pe.kind = 1;
          HIPCHK(h, hipEventRecord(pe.a, s));
        }
        HIPCHK(h, ut::launch_conv_c32s2(sl, s));
        if (h->profiling) {
          HIPCHK(h, hipEventRecord(pe.b, s));
          h->prof.push_back(pe);
        }
        return run_conv(h, b.conv2, tmp, dsbuf, y, n_img, H / 2, W / 2, true, false, s, sl.out1_max, y_max, mid_tid);
      }
    }
  }
  unsigned* tmp_max = nullptr;
  if ((rc = run_conv(h, b.conv1, x, nullptr, tmp, n_img, H, W, true, false, s, x_max, &tmp_max, x_tid))) return rc;
  const int Ho = (H + 2 - 3) / b.conv1.stride + 1, Wo = (W + 2 - 3) / b.conv1.stride + 1;
  const float* res = x;
  if (b.has_ds) {
    if ((rc = run_conv(h, b.ds, x, nullptr, dsbuf, n_img, H, W, false, false, s))) return rc;
    res = dsbuf;
  }
  return run_conv(h, b.conv2, tmp, res, y, n_img, Ho, Wo, true, false, s, tmp_max, y_max, mid_tid);
}

}  // namespace

extern "C" {

size_t ut_weight_blob_floats(void) { return UT_WEIGHT_BLOB_FLOATS; }

const char* ut_last_error(ut_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ut_canonical_backbone_weights(const float* blob, size_t n_floats, float* out, size_t out_floats, size_t* n_out) {
  if (!blob || !n_out) return fail(nullptr, UT_E_INVALID, "ut_canonical_backbone_weights: null argument");
  if (n_floats != UT_WEIGHT_BLOB_FLOATS) return fail(nullptr, UT_E_WEIGHTS, "ut_canonical_backbone_weights: weight blob has the wrong length");
  Cursor c{blob, n_floats};
  FoldedBackbone fbb;
  if (!fold_backbone(c, fbb)) return fail(nullptr, UT_E_WEIGHTS, "weight blob too short");
  std::vector<const Folded*> all = {&fbb.stem};
  for (const FoldedBlock& b : fbb.fb) {
    all.push_back(&b.conv1); all.push_back(&b.conv2);
    if (b.has_ds) all.push_back(&b.ds);
  }
  all.push_back(&fbb.proj);
  size_t n = 0;
  for (const Folded* f : all) n += f->w.size() + f->b.size();
  *n_out = n;
  if (!out) return UT_OK;
  if (out_floats < n) return fail(nullptr, UT_E_INVALID, "ut_canonical_backbone_weights: output too small");
  for (const Folded* f : all) {
    memcpy(out, f->w.data(), f->w.size() * sizeof(float)); out += f->w.size();
    memcpy(out, f->b.data(), f->b.size() * sizeof(float)); out += f->b.size();
  }
  return UT_OK;
}

int ut_create(int device, const float* blob, size_t n_floats, ut_handle* out) {
  if (!blob || !out) return fail(nullptr, UT_E_INVALID, "ut_create: null argument");
  if (n_floats != UT_WEIGHT_BLOB_FLOATS) return fail(nullptr, UT_E_WEIGHTS, "ut_create: weight blob has the wrong length");
  DeviceScope scope(device);          // the caller's current device is restored on return
  if (scope.err != hipSuccess) return fail(nullptr, UT_E_HIP, "hipSetDevice", scope.err);
  ut_handle h = new ut_context();
  h->device = device;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->num_cu = cus;
  }
  Cursor c{blob, n_floats};
  int rc = UT_OK;
  do {
    { float* cnt = nullptr; if ((rc = dev_alloc(h, &cnt, 2 * kMaxCounters + 64))) break; h->counters = (unsigned*)cnt;
      h->calib = h->counters + 2 * kMaxCounters;
      hipError_t e1 = hipMemset(h->calib, 0, 64 * sizeof(unsigned));
      if (e1 != hipSuccess) { rc = fail(h, UT_E_HIP, "calibration words", e1); break; } }
    { float* st = nullptr; if ((rc = dev_alloc(h, &st, 2))) break; h->status = (int*)st;
      hipError_t e2 = hipMemset(h->status, 0, 2 * sizeof(int));
      if (e2 == hipSuccess) e2 = hipHostMalloc((void**)&h->status_host, 2 * sizeof(int), hipHostMallocDefault);
      if (e2 != hipSuccess) { rc = fail(h, UT_E_HIP, "status words", e2); break; } }
    FoldedBackbone fbb;
    if (!fold_backbone(c, fbb)) { rc = fail(h, UT_E_WEIGHTS, "weight blob too short"); break; }
    const Folded& stem = fbb.stem;
    const Folded& proj = fbb.proj;
    const FoldedBlock* fb = fbb.fb;
    if ((rc = upload(h, stem.w, &h->stem_w)) || (rc = upload(h, stem.b, &h->stem_b))) break;
    for (int b = 0; b < 12 && !rc; ++b) rc = pack_block(h, h->bb[b], fb[b]);
    if (rc) break;
    if ((rc = pack_conv(h, h->proj, proj, 1, 1, 72))) break;
    // fusion 144 -> 108 -> 72 -> 72 (lib/models/model_utils.py:141-163)
    { const float* w0 = c.take(108 * 144); const float* b0 = c.take(108); BN bn0 = take_bn(c, 108);
      const float* w1 = c.take(72 * 108); const float* b1 = c.take(72); BN bn1 = take_bn(c, 72);
      const float* w2 = c.take(72 * 72); const float* b2 = c.take(72);
      if (!c.ok) { rc = fail(h, UT_E_WEIGHTS, "weight blob too short"); break; }
      if ((rc = pack_conv(h, h->fus0, w0, b0, &bn0, 144, 108, 1, 1, 108)) ||
          (rc = pack_conv(h, h->fus1, w1, b1, &bn1, 108, 72, 1, 1, 72)) ||
          (rc = pack_conv(h, h->fus2, w2, b2, nullptr, 72, 72, 1, 1, 72))) break; }
    // temporal 90 -> 90 x3 on a 92-channel padded layout (lib/models/temporal.py:31-38)
    for (int i = 0; i < 3 && !rc; ++i) {
      const float* tw = c.take(90 * 90); const float* tb = c.take(90);
      if (!c.ok) { rc = fail(h, UT_E_WEIGHTS, "weight blob too short"); break; }
      rc = pack_conv(h, h->tmp[i], tw, tb, nullptr, 90, 90, 1, 1, 92);
    }
    if (rc) break;
    // skeleton encoder (lib/models/skeleton_encoder.py:36-41)
    { const float* lw = c.take(144 * 132); const float* lb = c.take(144); BN bn = take_bn(c, 4);
      if (!c.ok) { rc = fail(h, UT_E_WEIGHTS, "weight blob too short"); break; }
      std::vector<float> sc(4), sh(4);
      for (int k = 0; k < 4; ++k) {
        double s = (double)bn.g[k] / sqrt((double)bn.v[k] + 1e-5);
        sc[k] = (float)s; sh[k] = (float)((double)bn.b[k] - (double)bn.m[k] * s);
      }
      if ((rc = upload(h, std::vector<float>(lw, lw + 144 * 132), &h->skel_w)) ||
          (rc = upload(h, std::vector<float>(lb, lb + 144), &h->skel_b)) ||
          (rc = upload(h, sc, &h->skel_scale)) || (rc = upload(h, sh, &h->skel_shift))) break; }
    if ((rc = take_regressor(h, c, h->reg_k, 76, 62))) break;
    if ((rc = take_regressor(h, c, h->reg_u, 72, 63))) break;
    if (!c.ok || c.left != 0) { rc = fail(h, UT_E_WEIGHTS, "weight blob length does not match the architecture"); break; }
  } while (0);
  if (rc) {
    g_create_error = h->err;
    ut_destroy(h);
    return rc;
  }
  *out = h;
  return UT_OK;
}

int ut_destroy(ut_handle h) {
  if (!h) return UT_OK;
  DeviceScope scope(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->status_host) (void)hipHostFree(h->status_host);
  for (int i = 0; i < 2; ++i) {
    if (h->lane_stream[i]) (void)hipStreamDestroy(h->lane_stream[i]);
    if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (auto& pe : h->prof) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
  delete h;
  return UT_OK;
}

int ut_set_backbone_chunk(ut_handle h, int crops_per_pass) {
  if (!h || crops_per_pass < 0) return fail(h, UT_E_INVALID, "ut_set_backbone_chunk: bad argument");
  if (crops_per_pass > kMaxChunk) return fail(h, UT_E_INVALID, "ut_set_backbone_chunk: at most 7281 crops per pass (32-bit offsets)");
  h->chunk = crops_per_pass == 0 ? kDefaultChunk : crops_per_pass;
  return UT_OK;
}

int ut_reserve(ut_handle h, int max_crops, int max_samples, int max_slots) {
  if (!h) return UT_E_INVALID;
  ON_DEVICE_OF(h);
  int rc;
  int c = max_crops < h->chunk ? max_crops : h->chunk;
  if (c > 0 && (rc = ensure_backbone_ws(h, c))) return rc;
  int cb = max_crops < PHASE_B_MAX ? max_crops : PHASE_B_MAX;
  if (cb > 0 && (rc = ensure_phase_b_ws(h, cb))) return rc;
  if (max_samples > 0 && (rc = ensure_head_ws(h, max_samples, max_samples))) return rc;
  if (max_slots > 0 && (rc = ensure_slots(h, max_slots, 0))) return rc;
  const size_t crops_need = (size_t)(max_crops > 0 ? max_crops : 0) * 96 * 96 * sizeof(float);
  if (crops_need > h->crops_ws_bytes) {
    HIPCHK(h, hipDeviceSynchronize());
    dev_free(h, h->crops_ws);
    h->crops_ws = nullptr; h->crops_ws_bytes = 0;
    float* pnew = nullptr;
    if ((rc = dev_alloc(h, &pnew, crops_need / 4))) return rc;
    h->crops_ws = pnew; h->crops_ws_bytes = crops_need;
  }
  return UT_OK;
}

int ut_warp_crops(ut_handle h, const uint8_t* src, int n_src_images, int src_h, int src_w, const double* cam_params,
                  const double* crop_params, const int32_t* src_index, int n_crops, int remap_mode, float* out,
                  void* stream) {
  // stateless: h may be NULL (then the call runs on the caller's current device and always checks synchronously)
  if (n_crops == 0) return UT_OK;
  if (!src || !cam_params || !crop_params || !src_index || !out || n_crops < 0 || src_h <= 0 || src_w <= 0 ||
      n_src_images <= 0 || (remap_mode != UT_REMAP_CV2_FIXED && remap_mode != UT_REMAP_FLOAT))
    return fail(h, UT_E_INVALID, "ut_warp_crops: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int *st_dev = nullptr, *st_host = nullptr, mode = UT_CHECK_SYNC, dev = 0;
  if (h) { st_dev = h->status; st_host = h->status_host; mode = h->check_mode; dev = h->device; }
  else {
    DevStatus st;
    int rc = stateless_status(&dev, &st);
    if (rc) return rc;
    st_dev = st.dev; st_host = st.host;
  }
  DeviceScope scope(dev);
  if (scope.err != hipSuccess) return fail(h, UT_E_HIP, "hipSetDevice", scope.err);
  HIPCHK(h, ut::launch_warp(src, n_src_images, src_h, src_w, cam_params, crop_params, src_index, n_crops, remap_mode,
                            out, nullptr, st_dev, s));
  if (mode == UT_CHECK_SYNC) {
    int sticky = 0, call = 0, rc = read_status(h, st_dev, st_host, s, &sticky, &call);
    if (rc) return rc;
    if (sticky & ut::UT_STATUS_ERRORS) {
      char buf[256];
      snprintf(buf, sizeof buf, "ut_warp_crops: %s", status_message(sticky));
      return fail(h, UT_E_INVALID, buf);
    }
  }
  return UT_OK;
}

extern "C" int ut_warp_map(const double* cam_params, const double* crop_params, const int32_t* src_index, int n_src_images,
                           int n_crops, float* out_map, void* stream) {
  if (n_crops == 0) return UT_OK;
  if (!cam_params || !crop_params || !src_index || !out_map || n_crops < 0 || n_src_images <= 0)
    return fail(nullptr, UT_E_INVALID, "ut_warp_map: bad argument");
  HIPCHK(nullptr, ut::launch_warp_map(cam_params, crop_params, src_index, n_src_images, n_crops, out_map, (hipStream_t)stream));
  return UT_OK;
}

// stem launch; *out_max receives its max word when the call runs the split-fp16 kernels
static int run_stem(ut_handle h, const float* crops, const uint8_t* crops_u8, float* out, int n, hipStream_t st,
                    unsigned** out_max) {
  *out_max = nullptr;
  if (h->call_split && !h->latency_mode) {
    int word = 0, rc = next_launch_word(h, st, &word);
    if (rc) return rc;
    *out_max = h->counters + kMaxCounters + word;
  }
  if (crops_u8) HIPCHK(h, ut::launch_stem_u8(crops_u8, h->stem_w, h->stem_b, out, n, *out_max, st));
  else HIPCHK(h, ut::launch_stem(crops, h->stem_w, h->stem_b, out, n, *out_max, st));
  return UT_OK;
}

// stem .. projection over crops given as fp32 (crops) or as u8 grey levels (crops_u8)
// One sub-batch of n crops (workspace slices starting at crop `off`) through stem .. projection on stream st.
static int backbone_pass(ut_handle h, const float* crops, const uint8_t* crops_u8, int off, int n, float* feat,
                         hipStream_t st) {
  int rc;
  const size_t a48 = (size_t)off * 48 * 48 * 32, a24 = (size_t)off * 24 * 24 * 64, a12 = (size_t)off * 12 * 12 * 128;
  unsigned* xm = nullptr;            // max word of the running activation
  if ((rc = run_stem(h, crops, crops_u8, h->bufX + a48, n, st, &xm))) return rc;
  {
    float *x = h->bufX + a48, *y = h->bufY + a48;
    int hw = 48;
    for (int b = 0; b < 5; ++b) {
      float* dst = b == 4 ? h->bufL2 + a24 : y;
      if ((rc = run_block(h, h->bb[b], x, h->bufH + a48, h->bufD + a24, dst, n, hw, hw, st, xm, &xm, 2 * b, 2 * b + 1))) return rc;
      hw = (hw + 2 - 3) / h->bb[b].conv1.stride + 1;
      float* t = x; x = y; y = t;
    }
  }
  const float* x = h->bufL2 + a24;
  float *y = h->bufP + a12, *other = h->bufQ + a12;
  int hw = 24;
  for (int b = 5; b < 12; ++b) {
    if ((rc = run_block(h, h->bb[b], x, h->bufBH + a12, h->bufBD + a12, y, n, hw, hw, st, xm, &xm, 2 * b, 2 * b + 1))) return rc;
    hw = (hw + 2 - 3) / h->bb[b].conv1.stride + 1;
    x = y;
    float* t = y; y = other; other = t;
  }
  // projection 256 -> 72, written NCHW like the reference (lib/models/model_utils.py:134)
  return run_conv(h, h->proj, x, nullptr, feat, n, 6, 6, false, true, st);
}

// stem .. projection over crops given as fp32 (crops) or as u8 grey levels (crops_u8)
static int run_backbone(ut_handle h, const float* crops, const uint8_t* crops_u8, int n_crops, float* feat, hipStream_t s) {
  int rc;
  const int chunk = n_crops < h->chunk ? n_crops : h->chunk;
  if ((rc = ensure_backbone_ws(h, chunk))) return rc;
  const int pass_b = n_crops < PHASE_B_MAX ? n_crops : PHASE_B_MAX;
  if ((rc = ensure_phase_b_ws(h, pass_b))) return rc;
  if ((rc = begin_call(h, s))) return rc;
  // One arithmetic per call, for every eligible layer of it: the split-fp16 kernels when the batch fills the chip with
  // their 256-row tiles down to the 6x6 maps (512 crops = 72 row tiles x 2 column tiles at layer4), else exact fp32.
  h->call_split = h->conv_arith == UT_CONV_SPLIT_F16_ALWAYS || (h->conv_arith == UT_CONV_SPLIT_F16 && n_crops >= 2 * h->num_cu);
  // ---- two lanes: the batch fits one pass of both phases and is big enough for two full-chip half-batches.  Every
  // launch is a persistent grid that drains a tile queue; its last round leaves workgroup slots idle for up to a tile
  // time (70-140 us of a 1.3 ms launch).  Frames are independent, so the two halves run the same launch sequence on
  // two streams and each one's idle slots are taken by the other's workgroups.  Same kernels on the same crops: the
  // results are bit-identical to the single-stream order.  (Not while profiling: per-launch event times would overlap.)
  if (h->lanes == 2 && !h->profiling && n_crops <= chunk && n_crops <= pass_b && n_crops >= 1024) {
    if (!h->lane_stream[0]) {
      for (int i = 0; i < 2; ++i) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->lane_stream[i], hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
      }
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    }
    HIPCHK(h, hipEventRecord(h->ev_fork, s));
    const int n0 = n_crops / 2;
    for (int i = 0; i < 2; ++i) {
      const int off = i ? n0 : 0, n = i ? n_crops - n0 : n0;
      HIPCHK(h, hipStreamWaitEvent(h->lane_stream[i], h->ev_fork, 0));
      const size_t first = (size_t)off * 96 * 96;
      if ((rc = backbone_pass(h, crops ? crops + first : nullptr, crops_u8 ? crops_u8 + first : nullptr, off, n,
                              feat + (size_t)off * 72 * 36, h->lane_stream[i])))
        return rc;
      HIPCHK(h, hipEventRecord(h->ev_join[i], h->lane_stream[i]));
    }
    for (int i = 0; i < 2; ++i) HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[i], 0));
    return UT_OK;
  }
  for (int base = 0; base < n_crops; base += pass_b) {
    const int nb = n_crops - base < pass_b ? n_crops - base : pass_b;
    // ---- phase A: stem + layer1 (48x48x32) + layer2 (24x24x64), `chunk` crops per pass
    // the words of one pass (<= 2 x 22 + 16 launches) come from one zeroing, so that none of them is recycled while live
    if (h->counter_next + 256 > kMaxCounters && (rc = begin_call(h, s))) return rc;
    unsigned* l2_max = nullptr;      // max word of bufL2: the passes' words merged when phase A took more than one pass
    const unsigned l2_gen = h->word_gen;
    int passes = 0;
    for (int done = 0; done < nb; done += chunk, ++passes) {
      const int n = nb - done < chunk ? nb - done : chunk;
      const size_t first = (size_t)(base + done) * 96 * 96;
      unsigned* xm = nullptr;
      if ((rc = run_stem(h, crops ? crops + first : nullptr, crops_u8 ? crops_u8 + first : nullptr, h->bufX, n, s, &xm))) return rc;
      float *x = h->bufX, *y = h->bufY;
      int hw = 48;
      for (int b = 0; b < 5; ++b) {
        float* dst = b == 4 ? h->bufL2 + (size_t)done * 24 * 24 * 64 : y;
        if ((rc = run_block(h, h->bb[b], x, h->bufH, h->bufD, dst, n, hw, hw, s, xm, &xm, 2 * b, 2 * b + 1))) return rc;
        hw = (hw + 2 - 3) / h->bb[b].conv1.stride + 1;
        float* t = x; x = y; y = t;
      }
      if (passes == 0) l2_max = xm;
      else if (l2_max && xm) HIPCHK(h, ut::launch_merge_max(l2_max, xm, s));
      else l2_max = nullptr;
    }
    // ---- phase B: layer3 (12x12x128) + layer4 (6x6x256) + projection over the whole pass
    const float* x = h->bufL2;
    float *y = h->bufP, *other = h->bufQ;
    int hw = 24;
    unsigned* xm = l2_gen == h->word_gen ? l2_max : nullptr;
    for (int b = 5; b < 12; ++b) {
      if ((rc = run_block(h, h->bb[b], x, h->bufBH, h->bufBD, y, nb, hw, hw, s, xm, &xm, 2 * b, 2 * b + 1))) return rc;
      hw = (hw + 2 - 3) / h->bb[b].conv1.stride + 1;
      x = y;
      float* t = y; y = other; other = t;
    }
    // projection 256 -> 72, written NCHW like the reference (lib/models/model_utils.py:134)
    if ((rc = run_conv(h, h->proj, x, nullptr, feat + (size_t)base * 72 * 36, nb, 6, 6, false, true, s))) return rc;
  }
  return UT_OK;
}

static int run_head(ut_handle h, const ut::HeadArgs& a, const float* skel, int n_skel, int mode, float* out_pose, float* out_raw,
                    hipStream_t s);

// Calibration of the regressor's four tensors (per regress mode): the head on the calibration crops' features, paired into two-view
// samples with canned cameras (f = 130 px, the second view 6 cm to the side), zero temporal memory in scratch state, a zero skeleton.
static int calibrate_head(ut_handle h, const float* feat, int n_crops, hipStream_t s) {
  const int S = n_crops / 2;
  int rc = ensure_head_ws(h, S, 1);
  if (rc) return rc;
  // one scratch allocation: [intrinsics 2S x 9 | extrinsics 2S x 16 | skeleton 132 | mem S x 648 | prev_ext S x 16 | pose S x 60 | raw S x 64]
  // floats, then [sample_range 2S | memory_idx S | hand_idx S] int64, [slot_seen S] int32, [use_memory S] bytes
  const size_t nf = (size_t)2 * S * 25 + 132 + (size_t)S * (648 + 16 + 60 + 64);
  const size_t off_i64 = (nf * 4 + 7) / 8 * 8, off_i32 = off_i64 + (size_t)4 * S * 8, off_u8 = off_i32 + (size_t)S * 4, total = off_u8 + S;
  std::vector<char> host(total, 0);
  float* f = reinterpret_cast<float*>(host.data());
  for (int c = 0; c < 2 * S; ++c) {
    float* k = f + (size_t)c * 9;
    k[0] = k[4] = 130.f; k[2] = k[5] = 47.5f; k[8] = 1.f;
    float* x = f + (size_t)2 * S * 9 + (size_t)c * 16;
    x[0] = x[5] = x[10] = x[15] = 1.f;
    if (c & 1) x[3] = 0.06f;
  }
  long long* i64 = reinterpret_cast<long long*>(host.data() + off_i64);
  for (int k = 0; k < S; ++k) { i64[2 * k] = 2 * k; i64[2 * k + 1] = 2 * k + 2; i64[2 * S + k] = k; i64[3 * S + k] = k & 1; }
  void* dv = nullptr;
  HIPCHK(h, hipMalloc(&dv, total));
  hipError_t e = hipMemcpyAsync(dv, host.data(), total, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { (void)hipFree(dv); return fail(h, UT_E_HIP, "calibration descriptors", e); }
  char* d = static_cast<char*>(dv);
  float* df = reinterpret_cast<float*>(d);
  ut::HeadArgs a{};
  a.feat = feat; a.intrinsics = df; a.extrinsics = df + (size_t)2 * S * 9;
  const float* skel = df + (size_t)2 * S * 25;
  a.mem = df + (size_t)2 * S * 25 + 132; a.prev_ext = a.mem + (size_t)S * 648;
  float* pose = a.prev_ext + (size_t)S * 16;
  float* raw = pose + (size_t)S * 60;
  a.sample_range = reinterpret_cast<const int64_t*>(d + off_i64); a.memory_idx = a.sample_range + 2 * S; a.hand_idx = a.memory_idx + S;
  a.slot_seen = reinterpret_cast<int*>(d + off_i32); a.use_memory = reinterpret_cast<const uint8_t*>(d + off_u8);
  a.n_samples = S; a.n_crops = 2 * S; a.n_slots = S; a.status = h->status; a.call_error_mask = 0;
  rc = begin_call(h, s);
  for (int mode = 0; mode < 2 && !rc; ++mode) rc = run_head(h, a, skel, 1, mode == 0 ? UT_MODE_KNOWN_SKELETON : UT_MODE_UNKNOWN_SKELETON, pose, raw, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(dv);
  return rc;
}

// Calibration of the split-fp16 activation scales: `crops` (device fp32 [n,96,96]) through the backbone with the per-launch
// (dynamic) scales, every tensor's largest magnitude folded into its calibrated word, then raised by 2^kCalibHeadroom.
// Synchronous; the status words are left as they were (a calibration pass reports nothing: a non-finite activation ends up in
// the calibrated word and is flagged by the calls that use it).
static int calibrate_split(ut_handle h, const float* crops, int n, hipStream_t s) {
  float* feat = nullptr;
  int rc = dev_alloc(h, &feat, (size_t)n * 72 * 36);
  if (rc) return rc;
  int saved[2] = {0, 0};
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipMemcpy(saved, h->status, sizeof saved, hipMemcpyDeviceToHost));
  const int arith = h->conv_arith, lanes = h->lanes;
  const bool prof = h->profiling;
  h->conv_arith = UT_CONV_SPLIT_F16_ALWAYS; h->lanes = 1; h->profiling = false; h->calibrating = true;
  const bool fusion = h->block_fusion;
  rc = (int)ut::launch_zero_words(h->calib, 64, s) != 0 ? fail(h, UT_E_HIP, "launch_zero_words") : UT_OK;
  // both launch forms of layer1 / layer2's entry: the separate-launch form (ut_set_block_fusion(h, 0)) has two tensors more
  for (int form = 0; form < 2 && !rc; ++form) {
    h->block_fusion = form == 0;
    rc = run_backbone(h, crops, nullptr, n, feat, s);
  }
  h->block_fusion = fusion;
  const bool with_head = n >= 2;
  if (!rc && with_head) rc = calibrate_head(h, feat, n, s);
  h->conv_arith = arith; h->lanes = lanes; h->profiling = prof; h->calibrating = false;
  if (!rc) {
    HIPCHK(h, ut::launch_raise_words(h->calib, kScaleTensors, kCalibHeadroom, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipMemcpy(h->status, saved, sizeof saved, hipMemcpyHostToDevice));
    h->calibrated = true;
    h->head_calibrated = with_head;
  }
  (void)hipStreamSynchronize(s);
  dev_free(h, feat);
  return rc;
}

constexpr int kBuiltinCalibCrops = 64;

static int calibrate_builtin(ut_handle h) {
  float* crops = nullptr;
  int rc = dev_alloc(h, &crops, (size_t)kBuiltinCalibCrops * 96 * 96);
  if (rc) return rc;
  hipError_t e = ut::launch_calibration_crops(crops, kBuiltinCalibCrops, 0);
  rc = e != hipSuccess ? fail(h, UT_E_HIP, "launch_calibration_crops", e) : calibrate_split(h, crops, kBuiltinCalibCrops, 0);
  (void)hipDeviceSynchronize();
  dev_free(h, crops);
  return rc;
}

int ut_calibrate_split(ut_handle h, const float* crops, int n_crops, void* stream) {
  if (!h) return UT_E_INVALID;
  if (n_crops < 0 || (n_crops > 0 && !crops)) return fail(h, UT_E_INVALID, "ut_calibrate_split: bad argument");
  ON_DEVICE_OF(h);
  return n_crops == 0 ? calibrate_builtin(h) : calibrate_split(h, crops, n_crops, (hipStream_t)stream);
}

int ut_set_split_scale(ut_handle h, int mode) {
  if (!h || (mode != UT_SPLIT_SCALE_CALIBRATED && mode != UT_SPLIT_SCALE_DYNAMIC)) return fail(h, UT_E_INVALID, "ut_set_split_scale: bad argument");
  ON_DEVICE_OF(h);
  h->scale_mode = mode;
  if (mode == UT_SPLIT_SCALE_CALIBRATED && h->conv_arith != UT_CONV_FP32 && !h->calibrated) return calibrate_builtin(h);
  return UT_OK;
}

int ut_get_split_calibration(ut_handle h, float* out33) {
  if (!h || !out33) return fail(h, UT_E_INVALID, "ut_get_split_calibration: null argument");
  ON_DEVICE_OF(h);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(out33, h->calib, kScaleTensors * sizeof(float), hipMemcpyDeviceToHost));
  return h->calibrated ? UT_OK : 1;
}

int ut_backbone(ut_handle h, const float* crops, int n_crops, float* feat, void* stream) {
  if (!h) return UT_E_INVALID;
  if (n_crops == 0) return UT_OK;
  if (!crops || !feat || n_crops < 0) return fail(h, UT_E_INVALID, "ut_backbone: bad argument");
  ON_DEVICE_OF(h);
  return run_backbone(h, crops, nullptr, n_crops, feat, (hipStream_t)stream);
}

int ut_warp_backbone(ut_handle h, const uint8_t* src, int n_src_images, int src_h, int src_w, const double* cam_params,
                     const double* crop_params, const int32_t* src_index, int n_crops, int remap_mode, float* feat,
                     void* stream) {
  if (!h) return UT_E_INVALID;
  if (n_crops == 0) return UT_OK;
  if (!src || !cam_params || !crop_params || !src_index || !feat || n_crops < 0 || src_h <= 0 || src_w <= 0 ||
      n_src_images <= 0 || (remap_mode != UT_REMAP_CV2_FIXED && remap_mode != UT_REMAP_FLOAT))
    return fail(h, UT_E_INVALID, "ut_warp_backbone: bad argument");
  ON_DEVICE_OF(h);
  hipStream_t s = (hipStream_t)stream;
  const bool u8 = remap_mode == UT_REMAP_CV2_FIXED;
  const size_t need = (size_t)n_crops * 96 * 96 * (u8 ? 1 : sizeof(float));
  if (need > h->crops_ws_bytes) {
    HIPCHK(h, hipDeviceSynchronize());
    dev_free(h, h->crops_ws);
    h->crops_ws = nullptr; h->crops_ws_bytes = 0;
    float* pnew = nullptr;
    int rc0 = dev_alloc(h, &pnew, (need + 3) / 4);
    if (rc0) return rc0;
    h->crops_ws = pnew; h->crops_ws_bytes = need;
  }
  HIPCHK(h, ut::launch_warp(src, n_src_images, src_h, src_w, cam_params, crop_params, src_index, n_crops, remap_mode,
                            u8 ? nullptr : (float*)h->crops_ws, u8 ? (uint8_t*)h->crops_ws : nullptr, h->status, s));
  if (h->check_mode == UT_CHECK_SYNC) {
    int sticky = 0, call = 0, rc = read_status(h, h->status, h->status_host, s, &sticky, &call);
    if (rc) return rc;
    if (sticky & ut::UT_STATUS_ERRORS) {
      char buf[256];
      snprintf(buf, sizeof buf, "ut_warp_backbone: %s", status_message(sticky));
      return fail(h, UT_E_INVALID, buf);
    }
  }
  return run_backbone(h, u8 ? nullptr : (const float*)h->crops_ws, u8 ? (const uint8_t*)h->crops_ws : nullptr, n_crops,
                      feat, s);
}

// The head behind the index checks: FTL, fusion, temporal block, regressor, decode (a.mem / a.prev_ext: the temporal state it
// reads and writes - the handle's, or scratch during a calibration pass).
// The regressor's four 3x3 convolutions (76 or 72 channels on the 6x6 map: 80 % of the head's arithmetic) run in the split-fp16
// arithmetic when the call is large enough to fill the chip with conv_w4's tiles: their tensors are then laid out with 128
// channels (zeros behind the real ones; zero weight rows and columns), the shape of layer3 at a 6x6 map.
static int run_head(ut_handle h, const ut::HeadArgs& a, const float* skel, int n_skel, int mode, float* out_pose, float* out_raw,
                    hipStream_t s) {
  int rc;
  const ut::HeadBuffers& b = h->hb;
  const int S = a.n_samples;
  h->call_split = false;             // fusion and temporal block: 1x1 convolutions, on the fp32 matrix instruction
  HIPCHK(h, ut::launch_ftl_in(a, b, s));
  if ((rc = run_conv(h, h->fus0, b.cat144, nullptr, b.f108, S, 6, 6, true, false, s))) return rc;
  if ((rc = run_conv(h, h->fus1, b.f108, nullptr, b.f72a, S, 6, 6, true, false, s))) return rc;
  if ((rc = run_conv(h, h->fus2, b.f72a, nullptr, b.f72b, S, 6, 6, false, false, s))) return rc;
  HIPCHK(h, ut::launch_ftl_out_temporal_in(a, b, s));
  if ((rc = run_conv(h, h->tmp[0], b.t92a, nullptr, b.t92b, S, 6, 6, true, false, s))) return rc;
  if ((rc = run_conv(h, h->tmp[1], b.t92b, nullptr, b.t92a, S, 6, 6, true, false, s))) return rc;
  if ((rc = run_conv(h, h->tmp[2], b.t92a, nullptr, b.t92b, S, 6, 6, false, false, s))) return rc;
  const Regressor& reg = mode == UT_MODE_KNOWN_SKELETON ? h->reg_k : h->reg_u;
  if (mode == UT_MODE_KNOWN_SKELETON)
    HIPCHK(h, ut::launch_skeleton(skel, h->skel_w, h->skel_b, h->skel_scale, h->skel_shift, b.skel, n_skel, s));
  // split-fp16 regressor: chosen per call like the backbone's arithmetic (UT_CONV_SPLIT_F16: from 4 x CUs samples = one tile of
  // 8 samples per CU and wave set), with calibrated scales only once the head has been calibrated
  const bool head_split = !h->latency_mode && (h->calibrating || h->scale_mode == UT_SPLIT_SCALE_DYNAMIC || h->head_calibrated) &&
                          (h->conv_arith == UT_CONV_SPLIT_F16_ALWAYS || (h->conv_arith == UT_CONV_SPLIT_F16 && S >= 4 * h->num_cu));
  const int stride = head_split ? kRegSplitCh : reg.c;
  const int tid0 = mode == UT_MODE_KNOWN_SKELETON ? 25 : 29;
  unsigned* in_word = nullptr;
  if (head_split) {
    int word = 0;
    if (h->counter_next + 16 > kMaxCounters && (rc = begin_call(h, s))) return rc;
    if ((rc = next_launch_word(h, s, &word))) return rc;
    in_word = h->counters + kMaxCounters + word;
  }
  HIPCHK(h, ut::launch_temporal_out(a, b.t92b, b.skel, n_skel, b.regin, reg.c, stride, in_word, s));
  // two BasicBlocks on the 6x6 map (lib/models/model_utils.py:195-208)
  h->call_split = head_split;
  const Block* blocks = head_split ? reg.blocks_split : reg.blocks;
  unsigned* mid_word = nullptr;
  rc = run_block(h, blocks[0], b.regin, b.rega, nullptr, b.regb, S, 6, 6, s, in_word, &mid_word, head_split ? tid0 : -1, head_split ? tid0 + 1 : -1);
  if (!rc) rc = run_block(h, blocks[1], b.regb, b.rega, nullptr, b.regin, S, 6, 6, s, mid_word, nullptr, head_split ? tid0 + 2 : -1, head_split ? tid0 + 3 : -1);
  h->call_split = false;
  if (rc) return rc;
  HIPCHK(h, ut::launch_pool_decode(a, b.regin, reg.c, stride, reg.w_out, reg.b_out, reg.d, out_pose, out_raw, b.rega, s));
  return UT_OK;
}

int ut_fuse_temporal_regress(ut_handle h, const float* feat, const float* intrinsics, const float* extrinsics,
                             const int64_t* sample_range, const int64_t* memory_idx, const uint8_t* use_memory,
                             const int64_t* hand_idx, int n_crops, int n_samples, int n_slots, int all_multiview,
                             const float* skel, int n_skel, int mode, float* out_pose, float* out_raw, void* stream) {
  if (!h) return UT_E_INVALID;
  if (n_samples == 0) return UT_OK;
  if (!feat || !intrinsics || !extrinsics || !sample_range || !memory_idx || !use_memory || !hand_idx || !out_pose ||
      n_samples < 0 || n_crops < n_samples || n_crops > 2 * n_samples || n_slots <= 0)
    return fail(h, UT_E_INVALID, "ut_fuse_temporal_regress: bad argument");
  if (mode == UT_MODE_KNOWN_SKELETON) {
    if (!skel || (n_skel != 1 && n_skel != n_samples))
      return fail(h, UT_E_INVALID, "ut_fuse_temporal_regress: skeleton must have 1 or n_samples entries");
  } else if (mode == UT_MODE_UNKNOWN_SKELETON) {
    // lib/models/umetrack_model.py:224-229
    if (!all_multiview)
      return fail(h, UT_E_UNSUPPORTED, "Unsupported: found single-view samples when calibration scale");
    n_skel = 0;
  } else {
    return fail(h, UT_E_INVALID, "ut_fuse_temporal_regress: unknown mode");
  }
  ON_DEVICE_OF(h);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure_head_ws(h, n_samples, n_skel))) return rc;
  if ((rc = ensure_slots(h, n_slots, s))) return rc;
  if ((rc = begin_call(h, s))) return rc;
  h->call_split = false;             // the head stays on the fp32 matrix instructions
  ut::HeadArgs a{};
  a.feat = feat; a.intrinsics = intrinsics; a.extrinsics = extrinsics; a.sample_range = sample_range;
  a.memory_idx = memory_idx; a.use_memory = use_memory; a.hand_idx = hand_idx; a.n_samples = n_samples;
  a.n_crops = n_crops; a.n_slots = n_slots;
  a.mem = h->mem; a.prev_ext = h->prev_ext;
  a.status = h->status; a.slot_seen = h->slot_seen;
  a.call_error_mask = mode == UT_MODE_UNKNOWN_SKELETON ? ut::UT_SINGLE_VIEW : 0;
  // index checks (stream ordered, in front of everything that indexes with the descriptors)
  HIPCHK(h, ut::launch_zero_words(h->status + 1, 1, s));
  HIPCHK(h, ut::launch_zero_words(h->slot_seen, (size_t)n_slots, s));
  HIPCHK(h, ut::launch_validate_desc(a, s));
  if (h->check_mode == UT_CHECK_SYNC) {
    int sticky = 0, call = 0;
    if ((rc = read_status(h, h->status, h->status_host, s, &sticky, &call))) return rc;
    if (sticky & ut::UT_STATUS_ERRORS) {
      char buf[256];
      snprintf(buf, sizeof buf, "ut_fuse_temporal_regress: %s", status_message(sticky));
      return fail(h, UT_E_INVALID, buf);
    }
    if (call & a.call_error_mask)   // lib/models/umetrack_model.py:224-229
      return fail(h, UT_E_UNSUPPORTED, "Unsupported: found single-view samples when calibration scale");
  }
  if (n_slots > h->slots_used) h->slots_used = n_slots;
  return run_head(h, a, skel, n_skel, mode, out_pose, out_raw, s);
}

int ut_reset_memory(ut_handle h) {
  if (!h) return UT_E_INVALID;
  ON_DEVICE_OF(h);
  HIPCHK(h, hipDeviceSynchronize());
  if (h->slots_cap) {
    HIPCHK(h, hipMemset(h->mem, 0, (size_t)h->slots_cap * 36 * 18 * sizeof(float)));
    HIPCHK(h, hipMemset(h->prev_ext, 0, (size_t)h->slots_cap * 16 * sizeof(float)));
  }
  h->slots_used = 0;
  return UT_OK;
}

int ut_get_memory(ut_handle h, float* mem, float* prev_ext, int max_slots, void* stream) {
  if (!h) return UT_E_INVALID;
  ON_DEVICE_OF(h);
  int n = h->slots_used < max_slots ? h->slots_used : max_slots;
  if (n > 0) {
    if (!mem || !prev_ext) return fail(h, UT_E_INVALID, "ut_get_memory: null output");
    HIPCHK(h, ut::launch_mem_export(h->mem, mem, n, (hipStream_t)stream));
    HIPCHK(h, hipMemcpyAsync(prev_ext, h->prev_ext, (size_t)n * 16 * sizeof(float), hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
  }
  return h->slots_used;
}

int ut_fk(ut_handle h, const float* hand_model, int n_models, const float* joint_angles, int ja_stride,
          const float* wrist_xf, int xf_stride, const int64_t* mirror, float t_scale, int n, float* out, void* stream) {
  // stateless: h may be NULL (errors then land in the thread-local slot read by ut_last_error(NULL))
  if (n == 0) return UT_OK;
  if (!hand_model || !joint_angles || !wrist_xf || !out || n < 0 || (n_models != 1 && n_models != n) ||
      ja_stride < 22 || xf_stride < 16)
    return fail(h, UT_E_INVALID, "ut_fk: bad argument");
  ON_DEVICE_IF(h);
  HIPCHK(h, ut::launch_fk(hand_model, n_models, joint_angles, ja_stride, wrist_xf, xf_stride, mirror, t_scale, n, out,
                          (hipStream_t)stream));
  return UT_OK;
}

int ut_gen_crop_cameras(ut_handle h, const double* cam_params, const double* camera_angles, const float* hand_model,
                        const float* joint_limits, int n_models, const float* joint_angles, const float* wrist_xf,
                        const int32_t* frame_idx, const int64_t* hand_idx, int n, int n_cams, int max_views, int min_vis,
                        int src_w, int src_h, int crop_size, double focal_multiplier, double* crop_params,
                        float* intrinsics, float* extrinsics, int32_t* cam_index, int32_t* n_views, int32_t* status,
                        float* landmarks, void* stream) {
  if (n == 0) return UT_OK;
  if (!cam_params || !camera_angles || !hand_model || !joint_limits || !joint_angles || !wrist_xf || !frame_idx ||
      !hand_idx || !crop_params || !intrinsics || !extrinsics || !cam_index || !n_views || !status || n < 0 ||
      (n_models != 1 && n_models != n) || n_cams <= 0 || max_views <= 0 || src_w <= 0 || src_h <= 0 || crop_size <= 1)
    return fail(h, UT_E_INVALID, "ut_gen_crop_cameras: bad argument");
  ON_DEVICE_IF(h);
  ut::CropGenArgs g{};
  g.cam_params = cam_params; g.camera_angles = camera_angles; g.hand_model = hand_model; g.joint_limits = joint_limits;
  g.joint_angles = joint_angles; g.wrist_xf = wrist_xf; g.frame_idx = frame_idx; g.hand_idx = hand_idx;
  g.n = n; g.n_models = n_models; g.n_cams = n_cams; g.max_views = max_views; g.min_vis = min_vis;
  g.src_w = src_w; g.src_h = src_h; g.crop_size = crop_size; g.focal_multiplier = focal_multiplier;
  g.crop_params = crop_params; g.intrinsics = intrinsics; g.extrinsics = extrinsics; g.cam_index = cam_index;
  g.n_views = n_views; g.status = status; g.landmarks = landmarks;
  HIPCHK(h, ut::launch_cropgen(g, (hipStream_t)stream));
  return UT_OK;
}

int ut_gen_crop_matrices(ut_handle h, const float* orig_extrinsics, const float* orig_intrinsics, const float* crop_points,
                         const int64_t* hand_idx, int n_frames, int n_views, int n_pts, int crop_size,
                         double focal_multiplier, float* extrinsics_xf, float* new_intrinsics, float* resample_xf,
                         int32_t* status, void* stream) {
  if (n_frames == 0 || n_views == 0) return UT_OK;
  if (!orig_extrinsics || !orig_intrinsics || !crop_points || !hand_idx || !extrinsics_xf || !new_intrinsics ||
      !resample_xf || !status || n_frames < 0 || n_views < 0 || n_pts <= 0 || crop_size <= 1)
    return fail(h, UT_E_INVALID, "ut_gen_crop_matrices: bad argument");
  ON_DEVICE_IF(h);
  ut::CropMatArgs g{};
  g.orig_extrinsics = orig_extrinsics; g.orig_intrinsics = orig_intrinsics; g.crop_points = crop_points;
  g.hand_idx = hand_idx; g.n_frames = n_frames; g.n_views = n_views; g.n_pts = n_pts; g.crop_size = crop_size;
  g.focal_multiplier = focal_multiplier; g.extrinsics_xf = extrinsics_xf; g.new_intrinsics = new_intrinsics;
  g.resample_xf = resample_xf; g.status = status;
  HIPCHK(h, ut::launch_cropmat(g, (hipStream_t)stream));
  return UT_OK;
}

int ut_resample_homography(ut_handle h, const void* src, int src_is_f32, int n, int src_h, int src_w,
                           const float* resample_xf, int out_h, int out_w, float* out, void* stream) {
  if (n == 0) return UT_OK;
  if (!src || !resample_xf || !out || n < 0 || src_h < 2 || src_w < 2 || out_h <= 0 || out_w <= 0)
    return fail(h, UT_E_INVALID, "ut_resample_homography: bad argument");
  ON_DEVICE_IF(h);
  HIPCHK(h, ut::launch_resample_homography(src, src_is_f32, n, src_h, src_w, resample_xf, out_h, out_w, out,
                                           (hipStream_t)stream));
  return UT_OK;
}

int ut_keypoint_metrics(ut_handle h, const float* gt, const float* tracked, const uint8_t* valid, int n_hands, int n_frames,
                        double* err, double* acc, double* gt_acc, uint8_t* valid_acc, void* stream) {
  if (n_hands == 0 || n_frames == 0) return UT_OK;
  if (!gt || !tracked || !valid || !err || n_hands < 0 || n_frames < 0 ||
      (n_frames >= 3 && (!acc || !gt_acc || !valid_acc)))
    return fail(h, UT_E_INVALID, "ut_keypoint_metrics: bad argument");
  ON_DEVICE_IF(h);
  HIPCHK(h, ut::launch_keypoint_metrics(gt, tracked, valid, n_hands, n_frames, err, acc, gt_acc, valid_acc,
                                        (hipStream_t)stream));
  return UT_OK;
}

int ut_set_backbone_lanes(ut_handle h, int lanes) {
  if (!h || (lanes != 1 && lanes != 2)) return fail(h, UT_E_INVALID, "ut_set_backbone_lanes: 1 or 2");
  h->lanes = lanes;
  return UT_OK;
}

int ut_set_conv_arithmetic(ut_handle h, int mode) {
  if (!h || (mode != UT_CONV_FP32 && mode != UT_CONV_SPLIT_F16 && mode != UT_CONV_SPLIT_F16_ALWAYS)) return fail(h, UT_E_INVALID, "ut_set_conv_arithmetic: bad argument");
  h->conv_arith = mode;
  if (mode != UT_CONV_FP32 && h->scale_mode == UT_SPLIT_SCALE_CALIBRATED && !h->calibrated) {
    ON_DEVICE_OF(h);
    return calibrate_builtin(h);
  }
  return UT_OK;
}

int ut_set_block_fusion(ut_handle h, int on) {
  if (!h) return UT_E_INVALID;
  h->block_fusion = on != 0;
  return UT_OK;
}

int ut_set_resident_weights(ut_handle h, int on) {
  if (!h) return UT_E_INVALID;
  if (on < 0 || on > 6) return fail(h, UT_E_INVALID, "ut_set_resident_weights: bad argument");
  h->resident_weights = on;
  return UT_OK;
}

int ut_set_latency_mode(ut_handle h, int on) {
  if (!h) return UT_E_INVALID;
  ON_DEVICE_OF(h);
  if (on && !h->splitk_ws) {
    const size_t n = 1u << 20;          // 4 MB: S x M x cout of every few-crop layer of this network is 442 k floats
    int rc;
    if ((rc = dev_alloc(h, &h->splitk_ws, n)) || (rc = dev_alloc(h, &h->zero_bias, 256))) return rc;
    HIPCHK(h, hipMemset(h->zero_bias, 0, 256 * sizeof(float)));
    h->splitk_floats = n;
  }
  h->latency_mode = on != 0;
  return UT_OK;
}

int ut_set_index_checks(ut_handle h, int mode) {
  if (!h || (mode != UT_CHECK_SYNC && mode != UT_CHECK_DEFERRED)) return fail(h, UT_E_INVALID, "ut_set_index_checks: bad argument");
  h->check_mode = mode;
  return UT_OK;
}

int ut_poll_status(ut_handle h, void* stream) {
  if (!h) return UT_E_INVALID;
  ON_DEVICE_OF(h);
  int sticky = 0, call = 0, rc = read_status(h, h->status, h->status_host, (hipStream_t)stream, &sticky, &call);
  if (rc) return rc;
  if (sticky & ut::UT_STATUS_ERRORS) {
    char buf[256];
    snprintf(buf, sizeof buf, "reported late (deferred checks): %s", status_message(sticky));
    return fail(h, UT_E_INVALID, buf);
  }
  return UT_OK;
}

int ut_status_snapshot(ut_handle h, int32_t* dst, void* stream) {
  if (!h || !dst) return fail(h, UT_E_INVALID, "ut_status_snapshot: null argument");
  ON_DEVICE_OF(h);
  HIPCHK(h, hipMemcpyAsync(dst, h->status, 2 * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return UT_OK;
}

int ut_profile_begin(ut_handle h, void* stream) {
  if (!h) return UT_E_INVALID;
  (void)stream;
  for (auto& pe : h->prof) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
  h->prof.clear();
  h->profiling = true;
  return UT_OK;
}

int ut_profile_end(ut_handle h, void* stream, double* conv_ms_total, int64_t* conv_launches, double* conv_flops_total) {
  if (!h) return UT_E_INVALID;
  ON_DEVICE_OF(h);
  h->profiling = false;
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  double ms = 0, fl = 0;
  for (auto& pe : h->prof) {
    float t = 0;
    HIPCHK(h, hipEventElapsedTime(&t, pe.a, pe.b));
    ms += t; fl += pe.flops;
  }
  if (conv_ms_total) *conv_ms_total = ms;
  if (conv_launches) *conv_launches = (int64_t)h->prof.size();
  if (conv_flops_total) *conv_flops_total = fl;
  for (auto& pe : h->prof) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
  h->prof.clear();
  return UT_OK;
}

int ut_profile_end_by_kind(ut_handle h, void* stream, double* ms2, int64_t* launches2, double* flops2) {
  if (!h || !ms2 || !launches2 || !flops2) return UT_E_INVALID;
  ON_DEVICE_OF(h);
  h->profiling = false;
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  for (int k = 0; k < 2; ++k) { ms2[k] = 0; launches2[k] = 0; flops2[k] = 0; }
  for (auto& pe : h->prof) {
    float t = 0;
    HIPCHK(h, hipEventElapsedTime(&t, pe.a, pe.b));
    const int k = pe.kind ? 1 : 0;
    ms2[k] += t; flops2[k] += pe.flops; launches2[k] += 1;
  }
  for (auto& pe : h->prof) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
  h->prof.clear();
  return UT_OK;
}

}  // extern "C"
