// Internal launch interface between the host orchestration (ut_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ut {

// One convolution (1x1 or 3x3, stride 1 or 2) as an implicit GEMM over NHWC activations:
//   out[m][n] = act( sum_k A[m][k] * Wp[n][k] + bias[n] (+ res[m][n]) ),  m = (img, oy, ox)
// k is ordered (channel slice, tap, channel in slice) with slices of `cslice` channels:
//   k = s*(taps*cslice) + tap*cslice + c ,  input channel = s*cslice + c
// cslice = 32 when cin % 32 == 0 (one K chunk = one tap of one 128-byte channel slice, so the 9 taps
// of a slice re-read the same cache lines back to back and hit L1), else cslice = cin (tap major).
// Wp is [cout_pad][k_pad] k-contiguous, zero padded, BatchNorm folded.
struct ConvLaunch {
  const float* in;     // [n_img, H, W, cin]          (cin % 4 == 0)
  const float* w;      // [cout_pad][k_pad]
  const void* w_split; // optional: the two fp16 planes of (w * scale) in fragment order (conv_split.hip), or null
  float split_unscale; // 1 / scale of w_split
  int* status;         // split kernels: sticky status word, UT_SPLIT_RANGE is set when the input holds an infinity or a NaN
  const unsigned* in_max;   // split kernels: device word with the bits of the magnitude the activation scale is taken from (see
                            // split_act_scale): the handle's calibrated word of this tensor, or the word the producing kernel
                            // left in this call (publish_abs_max)
  const unsigned* in_obs;   // optional, with a calibrated in_max: the word the producer left in THIS call - an input beyond the
                            // calibrated range sets UT_SPLIT_RANGE
  unsigned* out_max;        // optional device word (zero before the launch): receives the bits of max |out|
  const float* bias;   // [cout_pad]
  const float* res;    // optional residual, same layout as out
  float* out;          // NHWC [n_img, Ho, Wo, cout_store] or NCHW [n_img, cout_store, Ho*Wo]
  int n_img, H, W, cin, Ho, Wo;
  int cout_store;      // channels written per pixel (== channel stride of out / res)
  int cout_pad;        // rows of Wp (multiple of 128)
  int k_total;         // taps * cin
  int cslice;          // channel slice width of the k order (32 or cin)
  int k_pad;           // row stride of Wp (multiple of 32)
  int ksize, stride, pad;
  int relu;
  int out_nchw;
  int device;          // HIP device the launch goes to (the dynamic-LDS attribute is set once per device)
  int num_cu;          // compute units of the device (persistent grid sizing)
  unsigned* tile_counter;   // device word, zero before the launch: dynamic tile queue of the persistent grid
  int no_resident;     // split kernels, A/B tests (ut_set_resident_weights): bit 0 = not conv_c64k.hip, bit 1 = not conv_w4.hip, bit 2 =
                       // not conv_w4.hip on the 24x24x64 maps (conv_c64k / the chunked conv_split kernels take those layers), bit 3 = not
                       // conv_w4.hip's phase-plane form on the stride-2 entries of layer3 / layer4 (the chunked gather kernel takes them)
  int splits;          // > 1 (latency mode): K is cut in `splits` equal chunk ranges, out = [splits][M][cout_store] slabs;
                       // 1 = latency mode without a split (prefers small tiles); 0 = throughput dispatch
};

hipError_t launch_conv_igemm(const ConvLaunch& c, hipStream_t s);
// latency mode: out = act(bias + res + sum of the split-K slabs, in slab order)
hipError_t launch_splitk_finish(const float* slabs, int n_splits, int m, int cout_store, const float* bias,
                                const float* res, float* out, int relu, hipStream_t s);
// the same convolution on the fp16 matrix cores from two-piece splits of both operands (conv_split.hip)
bool conv_split_applicable(const ConvLaunch& c);
hipError_t launch_conv_split(const ConvLaunch& c, hipStream_t s);
// the stride-2 entry of layer2: BasicBlock's first convolution (3x3 / 2, 32 -> 64, BN, ReLU) and its shortcut (1x1 / 2, 32 -> 64, BN)
// from one pass over the input (conv_c32s2.hip; split-fp16 arithmetic, weights resident in registers)
struct Stride2Launch {
  const float* in;          // [n_img][H][W][32]
  float* out1;              // [n_img][H/2][W/2][64]: relu(bn1(conv1 x))
  float* out2;              // [n_img][H/2][W/2][64]: bn_d(downsample x)
  const void* w1_split;     // pack_split_weights planes of the 3x3 ([cout_pad / 32][9][...]) and of the 1x1 ([cout_pad / 32][1][...])
  const void* wd_split;
  float unscale1, unscale_d;      // 1 / (their weight scales)
  const float* bias1;
  const float* bias_d;
  const unsigned* in_max;   // scale word of `in` (see ConvLaunch::in_max / in_obs)
  const unsigned* in_obs;
  unsigned* out1_max;       // device word (zero before the launch): receives the bits of max |out1|
  int* status;
  int n_img, H, W;
  int device, num_cu;
};
bool conv_c32s2_applicable(const Stride2Launch& c);
hipError_t launch_conv_c32s2(const Stride2Launch& c, hipStream_t s);
// 3x3 stride-1 from >= 64 channels to a multiple of 128 channels: four waves of 128 x 64, weights straight from global memory into
// registers, the patch split on its way into LDS, one barrier per slice (conv_w4.hip); conv_split_kernel<256, 128, 4, 2, true>'s bits
bool conv_w4_applicable(const ConvLaunch& c);
hipError_t launch_conv_w4(const ConvLaunch& c, hipStream_t s);
// 3x3 stride-1 64 -> 64 channels with the weights resident in registers, K split across the two waves of a SIMD (conv_c64k.hip)
bool conv_c64k_applicable(const ConvLaunch& c);
hipError_t launch_conv_c64k(const ConvLaunch& c, hipStream_t s);
size_t pack_split_weights(const float* w, int cout_pad, int k_pad, float scale, uint16_t* out);
float split_weight_scale(const float* w, size_t n);
// One 32 -> 32 -> 32 channel BasicBlock (stride 1, no shortcut convolution) in one launch, split-fp16 arithmetic
// (conv_block32.hip): out = relu(conv2(relu(conv1(in) + bias1)) + bias2 + in), BatchNorm folded into weights and biases.
struct BlockLaunch {
  const float* in;          // [n_img, H, W, 32]
  float* out;               // [n_img, H, W, 32]
  const void* w1_split;     // ConvW::w_split of the two convolutions (fp16 planes in fragment order, pre-scaled)
  const void* w2_split;
  float unscale_w1, unscale_w2;   // 1 / weight scale
  const float* bias1;
  const float* bias2;
  float wsum1;              // max over output channels of sum_k |w1| (folded): bounds the intermediate with max|in| and bmax1
  float bmax1;              // max |bias1|
  const unsigned* in_max;   // scale word of `in` (never null; see ConvLaunch::in_max / in_obs)
  const unsigned* in_obs;
  unsigned* out_max;        // optional: receives the bits of max |out|
  int* status;
  unsigned* tile_counter;
  int n_img, H, W, device, num_cu;
};
bool conv_block32_applicable(const BlockLaunch& b);
hipError_t launch_conv_block32(const BlockLaunch& b, hipStream_t s);

// n 32-bit words := 0, as a kernel.  NOT hipMemsetAsync: captured into a hipGraph, a memset node of 16 bytes or more fills with
// a stale pattern from the second replay on (ROCm 7.2; tools/diag/graph_memset.py) - a garbage tile-queue word then walks a
// persistent kernel through ~10^9 tickets, which is what hung the whole-path graph replay of round 2.
hipError_t launch_zero_words(void* p, size_t n_words, hipStream_t s);
// *dst = max(*dst, *src) on two max words (one thread): the activation of several passes read by one consumer
hipError_t launch_merge_max(unsigned* dst, const unsigned* src, hipStream_t s);
// calibration: every non-zero finite word (the bits of a positive float) times 2^add_exp, exponent clamped below infinity
hipError_t launch_raise_words(unsigned* words, int n, int add_exp, hipStream_t s);
// n fp32 crops [n,96,96] of grey levels k / 255: the built-in calibration set of the split-fp16 activation scales (noise at
// several contrasts, gradients, bright blobs on a dark ground; counter-based, the same on every device)
hipError_t launch_calibration_crops(float* crops, int n, hipStream_t s);
// 3x3 stride-1 32->32 channel convs with the halo patch resident in LDS (conv_patch.hip)
bool conv_patch_applicable(const ConvLaunch& c);
hipError_t launch_conv_patch(const ConvLaunch& c, hipStream_t s);

// stem: conv3x3(1->32,pad 1)+BN+ReLU+maxpool2 ; crops [n,96,96] -> NHWC [n,48,48,32]
// out_max: optional device word (zero before the launch) that receives the bits of the largest output
hipError_t launch_stem(const float* crops, const float* w /*[32][9]*/, const float* bias /*[32]*/,
                       float* out, int n, unsigned* out_max, hipStream_t s);
// the same on the resampler's u8 grey levels (value / 255 on load)
hipError_t launch_stem_u8(const uint8_t* crops, const float* w, const float* bias, float* out, int n, unsigned* out_max,
                          hipStream_t s);

struct HeadBuffers {
  // workspace, all NHWC over the 6x6 map: [S,36,C]
  float* cat144;    // [S,36,144] canonical-space features of both views
  float* f108;      // [S,36,108]
  float* f72a;      // [S,36,72]
  float* f72b;      // [S,36,72] fusion output (canonical space)
  float* fused;     // [S,36,72] cam0-space fused features (input of the temporal block)
  float* t92a;      // [S,36,92] temporal ping
  float* t92b;      // [S,36,92] temporal pong
  float* regin;     // [S,36,C] regressor input (C = 76 or 72; 128 with zero channels when its convolutions run in split-fp16)
  float* rega;      // [S,36,C]
  float* regb;      // [S,36,C]
  float* skel;      // [n_skel,36,4]
  float* xf;        // [S,40] per-sample transforms: A0(12) A1(12) s0(1) rel(12) flags
};

// Bits of the sticky device status word written by the index checks (ut_api.hip turns them into UT_E_INVALID).
enum : int {
  UT_BAD_SAMPLE_RANGE = 1,   // a sample_range row is not 1 or 2 crops inside [0, n_crops]
  UT_BAD_MEMORY_IDX = 2,     // memory_idx outside [0, n_slots)
  UT_DUP_MEMORY_IDX = 4,     // two samples of one call name the same temporal slot
  UT_BAD_HAND_IDX = 8,       // hand_idx not 0 / 1
  UT_SINGLE_VIEW = 16,       // informational: at least one one-view sample (an error only in unknown-skeleton mode)
  UT_BAD_SRC_INDEX = 32,     // ut_warp_crops: src_index outside [0, n_src_images)
  UT_SPLIT_RANGE = 64,       // split-fp16 convolutions: an input activation is an infinity or a NaN (nothing to scale by)
};
constexpr int UT_STATUS_ERRORS = UT_BAD_SAMPLE_RANGE | UT_BAD_MEMORY_IDX | UT_DUP_MEMORY_IDX | UT_BAD_HAND_IDX | UT_BAD_SRC_INDEX | UT_SPLIT_RANGE;

#if defined(__HIPCC__)
// ---- activation range of the split-fp16 arithmetic -------------------------------------------------------------------
// Every kernel whose output a split convolution reads leaves the bits of max |out| in a device word (the values are
// compared as unsigned integers: for non-negative floats that is the float order, an infinity or a NaN compares above
// every finite value).  The consumer turns the word into a power of two 2^k that puts the largest activation in
// [2^14, 2^15), multiplies every activation by it before the two-piece split (exact in fp32) and folds 2^-k into the
// epilogue's factor: the first piece then never saturates and the second stays a normal fp16 number for every activation
// within 2^-18 of the layer's largest - the split has fp32's exponent range instead of fp16's, with no precondition on
// the network's activation magnitudes (lib/models/backbone_resnet.py:56-72 has none either).
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7FFFFFFFu; }

// one relaxed look first: after the first few workgroups the word is near its final value and the atomic is skipped
__device__ __forceinline__ void publish_abs_max(unsigned* word, unsigned lane_bits) {
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    const unsigned other = (unsigned)__shfl_xor((int)lane_bits, o);
    lane_bits = other > lane_bits ? other : lane_bits;
  }
  if ((threadIdx.x & 63) == 0 && lane_bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(word, lane_bits);
}

// scale = 2^k, unscale = 2^-k; ok = false when the word holds an infinity / a NaN (scale 1 then).
// in_obs (optional): the largest magnitude the producer actually stored in this call, when in_max is a calibrated word: the
// first piece of x * 2^k stays finite while x * 2^k < 2^16, i.e. while x's exponent is at most one above the word's;
// beyond that (or a non-finite value) ok = false.
__device__ __forceinline__ void split_act_scale(const unsigned* in_max, const unsigned* in_obs, float& scale, float& unscale, bool& ok) {
  const unsigned bits = (unsigned)__builtin_amdgcn_readfirstlane((int)*in_max);
  const int e = (int)(bits >> 23);
  ok = e != 255;
  int k = (bits == 0u || !ok) ? 0 : 141 - e;      // max in [2^(e-127), 2^(e-126)) -> times 2^k in [2^14, 2^15)
  k = k > 100 ? 100 : k < -100 ? -100 : k;
  scale = __uint_as_float((unsigned)(127 + k) << 23);
  unscale = __uint_as_float((unsigned)(127 - k) << 23);
  if (in_obs) {
    const int eo = (int)((unsigned)__builtin_amdgcn_readfirstlane((int)*in_obs) >> 23);
    if (eo == 255 || eo + k > 142) ok = false;      // 2^(eo - 127) * 2^k >= 2^16
  }
}
#endif

struct HeadArgs {
  const float* feat;        // [N,72,36] NCHW
  const float* intrinsics;  // [N,3,3]
  const float* extrinsics;  // [N,4,4]
  const int64_t* sample_range;
  const int64_t* memory_idx;
  const uint8_t* use_memory;
  const int64_t* hand_idx;
  int n_samples;
  int n_crops;
  int n_slots;
  float* mem;               // [slots,36,18] NHWC temporal memory
  float* prev_ext;          // [slots,16]
  int* status;              // device words: [0] sticky error bits (UT_* above), [1] bits of THIS call (zero on entry).
                            // Kernels that index with the descriptors return when status[0] has an error bit or
                            // status[1] a bit of call_error_mask: nothing is read or written out of range and the
                            // temporal state stays as it was
  int call_error_mask;      // UT_SINGLE_VIEW in unknown-skeleton mode (lib/models/umetrack_model.py:224-229), else 0
  int* slot_seen;           // [n_slots] scratch of the duplicate check, zero on entry
};

// One pass over the frame descriptors before anything indexes with them (stream ordered).
hipError_t launch_validate_desc(const HeadArgs& a, hipStream_t s);

hipError_t launch_ftl_in(const HeadArgs& a, const HeadBuffers& b, hipStream_t s);
hipError_t launch_ftl_out_temporal_in(const HeadArgs& a, const HeadBuffers& b, hipStream_t s);
// regin [S,36,reg_stride]: channels reg_c .. reg_stride - 1 are zeroed; out_max (optional, zero before the launch): max |regin|
hipError_t launch_temporal_out(const HeadArgs& a, const float* t_out /*[S,36,92]*/, const float* skel,
                               int n_skel, float* regin, int reg_c, int reg_stride, unsigned* out_max, hipStream_t s);
hipError_t launch_skeleton(const float* skel_in /*[n_skel,2,22,3]*/, const float* w /*[144][132]*/,
                           const float* bias /*[144]*/, const float* bn_scale /*[4]*/,
                           const float* bn_shift /*[4]*/, float* out /*[n_skel,36,4]*/, int n_skel,
                           hipStream_t s);
// avgpool(6x6) -> 1x1 conv (C->D) -> decode -> world transform -> pose record [S,60]
hipError_t launch_pool_decode(const HeadArgs& a, const float* reg_feat /*[S,36,reg_stride]*/, int reg_c, int reg_stride,
                              const float* w /*[D][C]*/, const float* bias /*[D]*/, int d,
                              float* out_pose, float* out_raw, float* raw_ws /*[S,64] scratch*/, hipStream_t s);

hipError_t launch_fk(const float* hand_model, int n_models, const float* ja, int ja_stride,
                     const float* xf, int xf_stride, const int64_t* mirror, float t_scale, int n,
                     float* out, hipStream_t s);

// Batched crop-camera generation (cropgen.hip): one candidate = one (frame, hand) label pose.
struct CropGenArgs {
  const double* cam_params;     // [n_frames*n_cams,32] source camera rows (layout of ut_warp_crops)
  const double* camera_angles;  // [n_cams] degrees
  const float* hand_model;      // [n_models,321]
  const float* joint_limits;    // [n_models,22,2]
  const float* joint_angles;    // [n,22]
  const float* wrist_xf;        // [n,4,4] (mm)
  const int32_t* frame_idx;     // [n]
  const int64_t* hand_idx;      // [n]
  int n, n_models, n_cams, max_views, min_vis, src_w, src_h, crop_size;
  double focal_multiplier;
  double* crop_params;          // [n,max_views,24]
  float* intrinsics;            // [n,max_views,3,3]
  float* extrinsics;            // [n,max_views,4,4]
  int32_t* cam_index;           // [n,max_views]  (-1 = unused slot)
  int32_t* n_views;             // [n]
  int32_t* status;              // [n] 0 ok, 1 = "Unable to create crop camera"
  float* landmarks;             // optional [n,21,3]: world landmarks of the label pose (mm)
};
hipError_t launch_cropgen(const CropGenArgs& g, hipStream_t s);

// torch_data path: crop matrices per (frame, view) and the pinhole->pinhole homography resampler.
struct CropMatArgs {
  const float* orig_extrinsics;  // [n_frames*n_views,4,4] world->eye
  const float* orig_intrinsics;  // [n_frames*n_views,3,3]
  const float* crop_points;      // [n_frames,n_pts,3]
  const int64_t* hand_idx;       // [n_frames] (1 = right hand = mirrored crop)
  int n_frames, n_views, n_pts, crop_size;
  double focal_multiplier;
  float* extrinsics_xf;          // [n_frames*n_views,4,4]
  float* new_intrinsics;         // [n_frames*n_views,3,3]
  float* resample_xf;            // [n_frames*n_views,4,4]
  int32_t* status;               // [n_frames*n_views]
};
hipError_t launch_cropmat(const CropMatArgs& g, hipStream_t s);
hipError_t launch_resample_homography(const void* src, int src_is_f32, int n, int src_h, int src_w,
                                      const float* resample_xf, int out_h, int out_w, float* out, hipStream_t s);

hipError_t launch_keypoint_metrics(const float* gt, const float* tracked, const uint8_t* valid, int n_hands, int n_frames,
                                   double* err, double* acc, double* gt_acc, uint8_t* valid_acc, hipStream_t s);

hipError_t launch_mem_export(const float* mem /*[slots,36,18]*/, float* out /*[slots,18,36]*/, int slots, hipStream_t s);

// status: device word; a crop whose src_index is outside [0, n_src) is written as zeros and sets UT_BAD_SRC_INDEX.
// out_u8 != NULL (mode 0 only): write the grey levels as u8 instead of out = level / 255.
hipError_t launch_warp(const uint8_t* src, int n_src, int src_h, int src_w, const double* cam,
                       const double* crop, const int32_t* src_index, int n_crops, int mode, float* out,
                       uint8_t* out_u8, int* status, hipStream_t s);

// diagnostic: the fp32 coordinate map of every crop, [n_crops, 96, 96, 2] (x, y); a bad src_index gives (-1, -1)
hipError_t launch_warp_map(const double* cam, const double* crop, const int32_t* src_index, int n_src, int n_crops,
                           float* out, hipStream_t s);

}  // namespace ut
