/* libumetrack_hip.so - C ABI of the MI355X-native UmeTrack per-frame inference hot path.
 *
 * The reference (2InfinityN6eyond/AbsoluteTrack) is pure Python: it has no FFI.  Each entry
 * point below replaces the Python/ATen/OpenCV code cited next to it; INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm tensors), fp32
 *    unless stated; index tensors are int64 like the reference's; the library owns only its
 *    handle, its packed weights, its activation workspace and the temporal-memory state.
 *  - every compute entry takes the hipStream_t to enqueue on (passed as void*); nothing
 *    synchronises the device except ut_create/ut_reserve/ut_destroy.
 *  - return value: 0 = ok, negative = error (UT_E_*); ut_last_error() gives the message.
 *  - a handle is bound to one device and is not thread-safe; distinct handles are independent
 *    (the reference runs one model per process, run_eval_known_skeleton.py:117-119).  Every entry
 *    that takes a handle runs on the handle's device whatever the caller's current device is and
 *    restores the caller's; entries called with handle == NULL run on the current device.
 *  - index tensors are checked ON THE DEVICE before anything indexes with them: src_index of
 *    ut_warp_crops, sample_range / memory_idx / hand_idx of ut_fuse_temporal_regress (the reference
 *    raises IndexError / asserts on these in Python: lib/tracker/tracker.py:330,
 *    lib/models/temporal.py:101-137, lib/models/umetrack_model.py:149-166,224-229).  A bad entry
 *    never leads to an out-of-range access and leaves the temporal state untouched.  By default
 *    (UT_CHECK_SYNC) the call reads the verdict back - one stream synchronisation - and returns
 *    UT_E_INVALID; with ut_set_index_checks(h, UT_CHECK_DEFERRED) nothing synchronises (needed inside
 *    hipGraph capture and for run-ahead launching), the affected work is skipped on the device and
 *    the error is reported by the next ut_poll_status.
 *  - stream capture: with UT_CHECK_DEFERRED and the workspace sized beforehand (ut_reserve, or one eager call of the same
 *    shapes) the compute entries only enqueue kernels on the given stream - no allocation, no synchronisation, and no memset
 *    nodes (the library zeroes its per-launch words with a kernel: captured hipMemsetAsync nodes replay a stale fill
 *    pattern under ROCm 7.2) - so ut_warp_backbone + ut_fuse_temporal_regress + ut_fk can be captured into ONE hipGraph
 *    and replayed (tests/test_gpu_tracker.py::test_whole_step_replays_from_one_hipgraph).
 */
#ifndef UMETRACK_HIP_H
#define UMETRACK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ut_context* ut_handle;

enum {
  UT_OK = 0,
  UT_E_INVALID = -1,   /* bad argument / shape */
  UT_E_HIP = -2,       /* a HIP runtime call failed */
  UT_E_WEIGHTS = -3,   /* weight blob has the wrong size */
  UT_E_UNSUPPORTED = -4
};

enum { UT_MODE_KNOWN_SKELETON = 0, UT_MODE_UNKNOWN_SKELETON = 1 };
enum { UT_CHECK_SYNC = 0, UT_CHECK_DEFERRED = 1 };
enum { UT_REMAP_CV2_FIXED = 0, UT_REMAP_FLOAT = 1 };

#define UT_CROP 96
#define UT_FEAT_CH 72
#define UT_FEAT_PIX 36
#define UT_POSE_REC 60        /* 22 joint angles | 16 wrist xf (row major, metres) | 1 scale | 21 sigmas */
#define UT_CAM_PARAMS 32      /* per source camera, doubles: see ut_warp_crops */
#define UT_CROP_PARAMS 24     /* per crop camera, doubles */
#define UT_HAND_MODEL_FLOATS 321
#define UT_WEIGHT_BLOB_FLOATS 4259410

/* Number of floats ut_create expects: the reference state_dict (lib/models/model_loader.py:84-87)
 * flattened in its own key order, every tensor as fp32 (num_batches_tracked as one float). */
size_t ut_weight_blob_floats(void);

/* lib/models/model_loader.py:53-88 (load_pretrained_model) + UmeTrackModel.to(device).
 * weights_blob is a HOST pointer.  BatchNorm is folded and the weights are packed for the
 * kernels here, once. */
int ut_create(int device, const float* weights_blob, size_t n_floats, ut_handle* out);

/* Host only (no device is touched): the backbone's convolutions as ut_create packs them - BatchNorm folded
 * (lib/models/backbone_resnet.py:56-72 in eval mode) and every inner channel brought to its canonical power-of-two scale:
 * output channel c of a producer (folded weight row and bias) times 2^k_c, input channel c of its consumers times 2^-k_c,
 * with 2^k_c putting the largest magnitude of the channel's defining rows into [1, 2).  Powers of two commute with ReLU and
 * with every fp32 rounding, so the packed network computes the checkpoint's fp32 values bit for bit, and two checkpoints that
 * differ by per-channel powers of two (a near-dead BatchNorm channel whose consumer weights compensate, say) pack to the SAME
 * tensors - the split-fp16 arithmetic, which keeps one power-of-two scale per tensor, therefore sees the same network.
 * out (may be NULL: only *n_out is set) receives, as fp32: stem w[32][1][9] b[32] | for each of the 12 BasicBlocks conv1
 * w[cout][cin][9] b[cout], conv2 w b, and the shortcut w[cout][cin][1] b where the block has one | projection w[72][256][1] b. */
int ut_canonical_backbone_weights(const float* weights_blob, size_t n_floats, float* out, size_t out_floats, size_t* n_out);
int ut_destroy(ut_handle h);
const char* ut_last_error(ut_handle h);   /* h may be NULL: error of the last failed ut_create */

/* UT_CHECK_SYNC (default) or UT_CHECK_DEFERRED, see "index tensors" above. */
int ut_set_index_checks(ut_handle h, int mode);
/* Synchronises `stream`, returns UT_E_INVALID (and clears the flag) if an index check has failed in any
 * call on this handle since the last poll, else UT_OK. */
int ut_poll_status(ut_handle h, void* stream);
/* Stream-ordered device-to-device copy of the two status words (dst: int32 [2] on the device; [0] != 0: a check has failed
 * since the last poll and the work that depended on it was skipped) - lets a UT_CHECK_DEFERRED caller that reads its
 * results back in one transfer see the verdict in that transfer and call ut_poll_status only when there is one. */
int ut_status_snapshot(ut_handle h, int32_t* dst, void* stream);

/* 1 (default) or 2: with 2, a ut_backbone / ut_warp_backbone call of >= 1024 crops that fits one workspace pass runs as
 * two half-batches on two internal streams (joined to the caller's stream before the call's work is visible to it), so
 * that the idle tail of one half's launches is filled by the other half's.  Same kernels on the same crops: results
 * are bit-identical (UT_CONV_FP32, and split-fp16 with calibrated scales; with UT_SPLIT_SCALE_DYNAMIC each half takes its own
 * scales).  Not applied between ut_profile_begin / ut_profile_end. */
int ut_set_backbone_lanes(ut_handle h, int lanes);

/* Arithmetic of the 3x3 convolutions of the backbone (all 24 of them: layer1 .. layer4; lib/models/backbone_resnet.py:56-72).
 *  UT_CONV_FP32        v_mfma_f32_32x32x2_f32: the exact fp32 multiply-add chain (default).
 *  UT_CONV_SPLIT_F16   both operands as two fp16 pieces (x0 = fp16(x), x1 = fp16(x - x0): 22 significand bits), three
 *                      piece products per k on v_mfma_f32_32x32x16_f16, fp32 accumulation: the terms dropped are ~2^-22
 *                      of a product, so the result carries fp32-level rounding error (not the fp32 chain's bits: outputs
 *                      agree with UT_CONV_FP32 to ~1e-6 relative) at up to 5.3x the matrix rate.
 *                      Range and scales.  Both operands are multiplied by exact powers of two before the split and the result
 *                      by the inverse: the weights by one power of two per layer, the activations by one per tensor - the
 *                      power that puts the tensor's scale word in [2^14, 2^15).  What that takes care of, and what it does not:
 *                      - a tensor's overall magnitude: any.  Networks whose activations are 2^-40 .. 2^+40 of another's give the
 *                        same bits, as with fp32;
 *                      - per-CHANNEL magnitudes: any.  ut_create packs every inner and trunk channel of the backbone at a
 *                        canonical power-of-two scale (see ut_canonical_backbone_weights), so a checkpoint's per-channel scale
 *                        freedom (near-dead BatchNorm channels with compensating consumer weights) never reaches the kernels;
 *                      - inside a packed tensor, a value more than 2^18 below the scale word (a weight that far below its layer's
 *                        largest) loses its second piece: it keeps an absolute error of 2^-40 of the scale word, where fp32
 *                        would keep 2^-24 of the value.  After canonicalisation such values belong to channels (weights) whose
 *                        contribution to the layer's output is below fp32's own rounding of that output;
 *                      - an infinity or a NaN has no scale, and (calibrated scales) an activation of 32 x the calibration maximum
 *                        or more would saturate the first piece: both set a sticky status bit that the next status read
 *                        (ut_poll_status, or any call that reads the index checks in UT_CHECK_SYNC mode) returns as
 *                        UT_E_INVALID "range check: ...".  Results of that call are then not to be used.
 *                      Where a tensor's scale word comes from: ut_set_split_scale.
 *                      The mode is chosen once per ut_backbone / ut_warp_backbone call, for every 3x3 convolution of it:
 *                      split for calls of >= 2 x (compute units) crops (512 on MI355X: their 256-row tiles then fill the
 *                      chip down to the 6x6 maps), exact fp32 below.
 *  UT_CONV_SPLIT_F16_ALWAYS  the same for calls of any size (slower on small ones: for tests).
 * The pose regressor's four 3x3 convolutions (lib/models/model_utils.py:195-208; 76 / 72 channels on the 6x6 map, run on tensors
 * zero-padded to 128 channels) follow the same choice per ut_fuse_temporal_regress call: split from 4 x (compute units) samples.
 * The stem, the 1x1 convolutions (shortcuts of layer3 / layer4, projection, fusion, temporal block) and every launch in latency
 * mode stay on the fp32 instruction in every mode. */
enum { UT_CONV_FP32 = 0, UT_CONV_SPLIT_F16 = 1, UT_CONV_SPLIT_F16_ALWAYS = 2 };
int ut_set_conv_arithmetic(ut_handle h, int mode);

/* Where the split-fp16 kernels take a tensor's power-of-two activation scale from.
 *  UT_SPLIT_SCALE_CALIBRATED (default)  one scale word per activation tensor of the backbone (25: the stem's output, every
 *      block's inner tensor and output) and of each pose regressor (4: its input, its blocks' inner tensors, the first block's
 *      output; calibrated on the calibration crops' features paired into two-view samples under canned cameras), fixed per handle: 2^4 x the tensor's largest magnitude over a calibration set.  A crop's
 *      result then does not depend on what else is in its batch: any batch size, pass size (ut_set_backbone_chunk), lane count or
 *      sharding of a frame set over ranks gives the same bits, as in UT_CONV_FP32 mode.  The calibration set is built in (64
 *      synthetic crops - noise at several contrasts, ramps, bright blobs on a dark ground - generated on the device, the same on
 *      every rank; run when split mode is first selected) or the caller's (ut_calibrate_split).  Every consumer compares the
 *      largest magnitude its producer stored in THIS call with the calibrated word: an input of 32 x the calibration maximum or
 *      more is reported as "range check" (see UT_CONV_SPLIT_F16), never silently saturated.
 *  UT_SPLIT_SCALE_DYNAMIC  the scale word is the one the producing kernel left in this call (the largest magnitude over the
 *      launch): adapts to any input, but a crop's low-order bits then depend on its batch (at the 1e-7 level). */
enum { UT_SPLIT_SCALE_CALIBRATED = 0, UT_SPLIT_SCALE_DYNAMIC = 1 };
int ut_set_split_scale(ut_handle h, int mode);
/* Replace the calibrated scale words by those of `crops` (device fp32 [n_crops,96,96], the tensor ut_backbone takes;
 * n_crops == 0: the built-in set).  Synchronous; results of later split-mode calls change at the 1e-7 level with it. */
int ut_calibrate_split(ut_handle h, const float* crops, int n_crops, void* stream);
/* The 33 calibrated scale words as floats (host pointer; 25 of the backbone, 2 x 4 of the two regressors); returns 1 when the
 * handle has not been calibrated yet. */
int ut_get_split_calibration(ut_handle h, float* out33);

/* Split-fp16 mode only: run each BasicBlock of layer1 (32 -> 32 -> 32 channels at 48x48) as ONE launch whose intermediate
 * relu(bn1(conv1 x)) stays in LDS (csrc/conv_block32.hip) instead of two convolution launches with a round trip through HBM
 * (1 = default).  Same arithmetic; the intermediate's power-of-two scale comes from a bound instead of the measured maximum,
 * so the two forms agree to the split arithmetic's rounding (~1e-7 relative), not bit for bit.  The same switch covers layer2's
 * entry (csrc/conv_c32s2.hip: the block's stride-2 3x3 convolution and its 1x1 shortcut as one launch instead of a split-fp16 and
 * an fp32 launch; the shortcut then runs in the split arithmetic too).  0 is for A/B tests. */
int ut_set_block_fusion(ut_handle h, int on);

/* Split-fp16 mode only: which kernels take the 3x3 convolutions of layer2 .. layer4 (A/B switch for tests; 1 = default).
 * csrc/conv_w4.hip: tiles of whole maps (288 pixels x 128 channels at 12x12 and 6x6, one 24x24 map x 64 channels), four waves of
 *    288 pixels x 32 channels, weights global -> registers, the patch split on its way into LDS at padded image coordinates, one
 *    barrier per slice.  Stride 1 at 12x12 / 6x6: the chunked kernel's bits; at 24x24 (16-channel slices) and on the stride-2
 *    entries of layer3 / layer4 (four phase planes of the input, summed plane by plane) its sum in another order: equal to fp32
 *    rounding, deterministic.
 * csrc/conv_c64k.hip (layer2's 64 -> 64, the form conv_w4 replaced): weights resident in registers, K split across the two waves
 *    of a SIMD; the chunked kernel's sum to fp32 rounding.
 * csrc/conv_split.hip: the chunked kernels (layer2's stride-2 entry is csrc/conv_c32s2.hip in any case).
 * 1: conv_w4 wherever it applies.  0: the chunked kernels everywhere.  6: as 1 with the stride-2 entries through the chunked gather
 * kernel.  With those through the gather kernel as well - 2: conv_c64k on 24x24, chunked elsewhere.  4: conv_w4 at 12x12 / 6x6,
 * conv_c64k at 24x24.  5: conv_w4 at 12x12 / 6x6, chunked at 24x24.  3: conv_w4 on the stride-1 convolutions of all three sizes. */
int ut_set_resident_weights(ut_handle h, int on);

/* Latency mode for calls on a handful of crops (the per-frame tracker): convolutions whose launch has far fewer tiles
 * than the chip has CUs split K across workgroups and add the partial sums in a fixed order.  Deterministic, but not
 * the unsplit kernel's summation order: results agree with the default mode to fp32 rounding (~1e-6 relative), not
 * bit for bit, which is why it is opt-in (0 = off, the default).  Large batches are unaffected. */
int ut_set_latency_mode(ut_handle h, int on);

/* Pre-size the activation workspace / temporal state so that later calls never allocate
 * (needed before capturing calls into a hipGraph). */
int ut_reserve(ut_handle h, int max_crops, int max_samples, int max_slots);

/* Crops processed per pass of the early (48x48, 24x24) backbone layers; 0 = library default
 * (4096), at most 7281 (32-bit buffer offsets).  Workspace grows with it: ~1 MB per crop. */
int ut_set_backbone_chunk(ut_handle h, int crops_per_pass);

/* lib/tracker/tracker.py:61-89 (_warp_image) + :332 (/255) for a batch of crops.
 *  src          u8 [n_src_images, src_h, src_w]
 *  cam_params   f64 [n_src_images, 32]: fx fy cx cy | k1 k2 k3 k4 p1 p2 k5 k6 | R(9, row major) t(3)
 *               of camera_to_world | 8 unused           (lib/common/camera.py:109-143,296-312)
 *  crop_params  f64 [n_crops, 24]: fx fy cx cy | R(9) t(3) of the crop camera_to_world | 8 unused
 *               (lib/common/camera.py:61-75,320-329)
 *  src_index    i32 [n_crops] which source image each crop samples
 *  out          f32 [n_crops, 96, 96] in [0,1]
 *  (stateless: h may be NULL; then the index check is always synchronous)
 *  remap_mode   UT_REMAP_CV2_FIXED: OpenCV's 8-bit INTER_LINEAR arithmetic (coordinates to 1/32
 *               px, 15-bit weights, rounded u8) then /255;  UT_REMAP_FLOAT: exact float bilinear. */
int ut_warp_crops(ut_handle h, const uint8_t* src, int n_src_images, int src_h, int src_w,
                  const double* cam_params, const double* crop_params, const int32_t* src_index,
                  int n_crops, int remap_mode, float* out, void* stream);

/* Diagnostic / test entry: the fp32 coordinate map ut_warp_crops samples with, i.e. the array the reference hands to
 * cv2.remap (lib/tracker/tracker.py:69-85: fp64 camera arithmetic cast to float32).  out_map f32 [n_crops,96,96,2] (x, y);
 * cam_params / crop_params / src_index as for ut_warp_crops; a src_index outside [0, n_src_images) gives (-1, -1).
 * Stateless, runs on the current device. */
int ut_warp_map(const double* cam_params, const double* crop_params, const int32_t* src_index, int n_src_images,
                int n_crops, float* out_map, void* stream);

/* FeatureExtractor._image_backbone: lib/models/model_utils.py:107-138,
 * lib/models/backbone_resnet.py:14-192, called at lib/models/umetrack_model.py:127-129.
 *  crops f32 [n_crops,96,96] -> feat f32 [n_crops,72,6,6] (NCHW like the reference). */
int ut_backbone(ut_handle h, const float* crops, int n_crops, float* feat, void* stream);

/* ut_warp_crops followed by ut_backbone without materialising the fp32 crop tensor: the resampled crops stay in
 * the handle's workspace - as the exact u8 grey levels OpenCV's 8-bit remap produces in UT_REMAP_CV2_FIXED mode
 * (lib/tracker/tracker.py:87 returns u8, :332 divides by 255: the stem applies the /255 on load, same bits), as
 * fp32 in UT_REMAP_FLOAT mode.  Arguments as for the two calls it replaces (lib/tracker/tracker.py:61-89,332 +
 * lib/models/umetrack_model.py:127-129). */
int ut_warp_backbone(ut_handle h, const uint8_t* src, int n_src_images, int src_h, int src_w,
                     const double* cam_params, const double* crop_params, const int32_t* src_index,
                     int n_crops, int remap_mode, float* feat, void* stream);

/* Everything after the backbone in UmeTrackModel.regress_pose_use_skeleton /
 * regress_pose_pred_skel_scale (lib/models/umetrack_model.py:131-242): single-view xfs, FTL,
 * 2-view fusion, temporal ConvRNN (state kept in the handle, lib/models/temporal.py:93-139),
 * skeleton encoder, pose regressor, decoders (lib/models/regressor.py:76-121,163-186),
 * Procrustes (lib/models/model_utils.py:17-54) and the world transform (:77-90).
 *  feat [n_crops,72,6,6]; intrinsics [n_crops,3,3]; extrinsics [n_crops,4,4];
 *  sample_range i64 [n_samples,2] (1 or 2 views per sample, inside [0,n_crops]); memory_idx i64
 *  [n_samples] (distinct slots, all in [0,n_slots)); use_memory u8 [n_samples]; hand_idx i64
 *  [n_samples] in {0,1} - all checked on the device, UT_E_INVALID otherwise;
 *  skel f32 [n_skel,2,22,3] = (joint_rotation_axes, joint_rest_positions[m]) with n_skel in
 *  {1, n_samples}, NULL in UT_MODE_UNKNOWN_SKELETON (single-view samples are then rejected with
 *  UT_E_UNSUPPORTED: on the host when `all_multiview` is 0, else by the per-sample device check);
 *  n_slots = max(memory_idx)+1 as computed by the caller (lib/models/temporal.py:102);
 *  out_pose f32 [n_samples,60]; out_raw (optional, may be NULL) f32 [n_samples,64] regressor
 *  output before decoding. */
int ut_fuse_temporal_regress(ut_handle h, const float* feat, const float* intrinsics,
                             const float* extrinsics, const int64_t* sample_range,
                             const int64_t* memory_idx, const uint8_t* use_memory,
                             const int64_t* hand_idx, int n_crops, int n_samples, int n_slots,
                             int all_multiview, const float* skel, int n_skel, int mode,
                             float* out_pose, float* out_raw, void* stream);

/* Drop the temporal state (a fresh SimpleConvRNN, lib/models/temporal.py:40-41). */
int ut_reset_memory(ut_handle h);
/* Copy out the state for inspection: mem f32 [n_slots,18,6,6] (NCHW), prev_ext f32 [n_slots,4,4];
 * returns the number of slots currently held (<= max_slots copied). */
int ut_get_memory(ut_handle h, float* mem, float* prev_ext, int max_slots, void* stream);

/* lib/common/hand_skinning.py:189-209 (skin_landmarks) for a batch of poses.
 *  hand_model f32 [n_models, 321]: axes[22,3] rest[22,3] landmark_rest[21,3] bone_weights[21,3]
 *             bone_indices[21,3] (as floats); n_models in {1, n}
 *  joint_angles [n,22]; wrist_xf [n,4,4] with row strides given in floats (so that the pose
 *  record of ut_fuse_temporal_regress can be consumed in place); translation is multiplied by
 *  t_scale (1000 for metres->mm, lib/tracker/tracker.py:379) and column 0 is negated where
 *  mirror[i] != 0 (lib/tracker/perspective_crop.py:48-49; mirror may be NULL);
 *  out [n,21,3].  Stateless: h may be NULL. */
int ut_fk(ut_handle h, const float* hand_model, int n_models, const float* joint_angles,
          int ja_stride, const float* wrist_xf, int xf_stride, const int64_t* mirror, float t_scale,
          int n, float* out, void* stream);

/* HandTracker.gen_crop_cameras for a batch of (frame, hand) label poses in one launch
 * (lib/tracker/tracker.py:222-260 -> lib/tracker/perspective_crop.py:136-180 -> lib/common/crop.py:31-82,
 * lib/common/affine.py:34-76) plus the network camera inputs of lib/tracker/tracker.py:333-337.
 *  cam_params    f64 [n_frames*n_cams,32]  source cameras, rows as for ut_warp_crops
 *  camera_angles f64 [n_cams]              roll of each camera in degrees
 *  hand_model    f32 [n_models,321], joint_limits f32 [n_models,22,2]; n_models in {1, n}
 *  joint_angles  f32 [n,22], wrist_xf f32 [n,4,4] (mm, world), frame_idx i32 [n], hand_idx i64 [n]
 *  Crop points are the landmarks of the pose, of the neutral pose (mid joint limits) and of the
 *  zero pose (num_crop_points = 63); cameras with >= min_vis of the pose's 21 landmarks inside the
 *  src_w x src_h image and in front are eligible, the first max_views of them in index order are
 *  used (sort_camera_index=True), right hands (hand_idx==1) get the x-mirrored crop.
 *  Outputs, padded to max_views per candidate:
 *  crop_params f64 [n,max_views,24] (rows as for ut_warp_crops), intrinsics f32 [n,max_views,3,3],
 *  extrinsics f32 [n,max_views,4,4] (world->eye, translation in metres), cam_index i32 [n,max_views]
 *  (-1 = unused), n_views i32 [n], status i32 [n] (1 where the reference raises "Unable to create
 *  crop camera", lib/common/crop.py:25-26); landmarks f32 [n,21,3] (optional, may be NULL): the world
 *  landmarks of each pose, i.e. landmarks_from_hand_pose(hand_model, pose, hand_idx) of
 *  lib/tracker/perspective_crop.py:40-51 (the crop points' first 21; same arithmetic as ut_fk).
 *  Stateless: h may be NULL. */
int ut_gen_crop_cameras(ut_handle h, const double* cam_params, const double* camera_angles,
                        const float* hand_model, const float* joint_limits, int n_models,
                        const float* joint_angles, const float* wrist_xf, const int32_t* frame_idx,
                        const int64_t* hand_idx, int n, int n_cams, int max_views, int min_vis,
                        int src_w, int src_h, int crop_size, double focal_multiplier,
                        double* crop_params, float* intrinsics, float* extrinsics,
                        int32_t* cam_index, int32_t* n_views, int32_t* status, float* landmarks,
                        void* stream);

/* torch_data path, lib/batched_dataset/data_transform.py:147-212 (_gen_crop_matrices) for every
 * (frame, view) of a batch in one launch.
 *  orig_extrinsics f32 [n_frames*n_views,4,4] world->eye of the (pinhole) source cameras
 *  orig_intrinsics f32 [n_frames*n_views,3,3] (fx, fy, cx, cy are read)
 *  crop_points     f32 [n_frames,n_pts,3]     points the crop must enclose (same units as the extrinsics)
 *  hand_idx        i64 [n_frames]             1 = right hand: x-mirrored crop
 *  Outputs: extrinsics_xf f32 [.,4,4] (world->eye of the crop camera), new_intrinsics f32 [.,3,3],
 *  resample_xf f32 [.,4,4] (crop pixel (u,v,1,1) -> source pixel, data_transform.py:57-76),
 *  status i32 [.] (1 where the reference raises "Unable to create crop camera").  h may be NULL. */
int ut_gen_crop_matrices(ut_handle h, const float* orig_extrinsics, const float* orig_intrinsics,
                         const float* crop_points, const int64_t* hand_idx, int n_frames, int n_views,
                         int n_pts, int crop_size, double focal_multiplier, float* extrinsics_xf,
                         float* new_intrinsics, float* resample_xf, int32_t* status, void* stream);

/* lib/batched_dataset/data_transform.py:79-144 (_resample_images_batched) + the /255 of :281.
 *  src: n images [n,src_h,src_w], u8 (src_is_f32 = 0) or f32 (1, the reference's astype(float32) copy)
 *  resample_xf f32 [n,4,4]; out f32 [n,out_h,out_w] = bilinear sample / 255 where the 2x2
 *  neighbourhood lies inside the source (0 <= x < src_w-1, 0 <= y < src_h-1), 0 elsewhere.
 *  h may be NULL. */
int ut_resample_homography(ut_handle h, const void* src, int src_is_f32, int n, int src_h, int src_w,
                           const float* resample_xf, int out_h, int out_w, float* out, void* stream);

/* Per-frame metrics of the eval scripts (load_eval.py:26-45, run_eval_known_skeleton.py:92-93).
 *  gt, tracked f32 [n_hands,n_frames,21,3]; valid u8 [n_hands,n_frames]
 *  err f64 [n_hands,n_frames]       mean over landmarks of |gt - tracked| (every frame; mask with `valid`)
 *  acc, gt_acc f64 [n_hands,n_frames-2]  mean |p[t] + p[t+2] - 2 p[t+1]| of tracked / gt
 *  valid_acc u8 [n_hands,n_frames-2]     valid[t] & valid[t+1] & valid[t+2]
 *  (acc outputs may be NULL only if n_frames < 3).  h may be NULL. */
int ut_keypoint_metrics(ut_handle h, const float* gt, const float* tracked, const uint8_t* valid, int n_hands,
                        int n_frames, double* err, double* acc, double* gt_acc, uint8_t* valid_acc,
                        void* stream);

/* Names of the kernels launched by the calls above and the average duration in ms of the
 * dominant (implicit-GEMM convolution) kernel measured with hipEvents on `stream` between
 * ut_profile_begin and ut_profile_end (bench.py roofline leg). */
int ut_profile_begin(ut_handle h, void* stream);
int ut_profile_end(ut_handle h, void* stream, double* conv_ms_total, int64_t* conv_launches,
                   double* conv_flops_total);
/* The same, separately for [0] the launches on the fp32 matrix instructions and [1] the split-fp16 launches
 * (ut_set_conv_arithmetic): three arrays of two. */
int ut_profile_end_by_kind(ut_handle h, void* stream, double* ms2, int64_t* launches2, double* flops2);

#ifdef __cplusplus
}
#endif
#endif
