"""ctypes binding of libumetrack_hip.so (include/umetrack_hip.h).

There is no CPU fallback: if the shared library is missing or no HIP device is
present, every entry point raises.  PyTorch-ROCm is used for device memory and
streams only; tensors cross the boundary as raw device pointers.
"""
import contextlib
import ctypes
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import arch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libumetrack_hip.so")

EXPORTS = (
    "ut_weight_blob_floats", "ut_create", "ut_destroy", "ut_last_error", "ut_reserve",
    "ut_set_backbone_chunk", "ut_warp_crops", "ut_backbone", "ut_fuse_temporal_regress",
    "ut_reset_memory", "ut_get_memory", "ut_fk", "ut_gen_crop_cameras", "ut_gen_crop_matrices",
    "ut_resample_homography", "ut_keypoint_metrics", "ut_profile_begin", "ut_profile_end", "ut_profile_end_by_kind",
    "ut_set_index_checks", "ut_poll_status", "ut_warp_backbone", "ut_set_latency_mode", "ut_set_conv_arithmetic",
    "ut_set_backbone_lanes", "ut_status_snapshot", "ut_warp_map", "ut_set_block_fusion", "ut_set_resident_weights",
    "ut_canonical_backbone_weights", "ut_set_split_scale", "ut_calibrate_split", "ut_get_split_calibration",
)

UT_MODE_KNOWN, UT_MODE_UNKNOWN = 0, 1
UT_REMAP_CV2_FIXED, UT_REMAP_FLOAT = 0, 1
UT_CHECK_SYNC, UT_CHECK_DEFERRED = 0, 1

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load_library() -> ctypes.CDLL:
    """dlopen the in-tree library and declare the prototypes.  Raises loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, f32p = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    lib.ut_weight_blob_floats.restype = ctypes.c_size_t
    lib.ut_weight_blob_floats.argtypes = []
    lib.ut_create.restype = i32
    lib.ut_create.argtypes = [i32, vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.ut_destroy.restype = i32
    lib.ut_destroy.argtypes = [vp]
    lib.ut_last_error.restype = ctypes.c_char_p
    lib.ut_last_error.argtypes = [vp]
    lib.ut_reserve.restype = i32
    lib.ut_reserve.argtypes = [vp, i32, i32, i32]
    lib.ut_set_backbone_chunk.restype = i32
    lib.ut_set_backbone_chunk.argtypes = [vp, i32]
    lib.ut_warp_crops.restype = i32
    lib.ut_warp_crops.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32, i32, f32p, vp]
    lib.ut_warp_backbone.restype = i32
    lib.ut_warp_backbone.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32, i32, f32p, vp]
    lib.ut_backbone.restype = i32
    lib.ut_backbone.argtypes = [vp, f32p, i32, f32p, vp]
    lib.ut_fuse_temporal_regress.restype = i32
    lib.ut_fuse_temporal_regress.argtypes = [vp, f32p, f32p, f32p, vp, vp, vp, vp, i32, i32, i32, i32,
                                             f32p, i32, i32, f32p, f32p, vp]
    lib.ut_reset_memory.restype = i32
    lib.ut_reset_memory.argtypes = [vp]
    lib.ut_get_memory.restype = i32
    lib.ut_get_memory.argtypes = [vp, f32p, f32p, i32, vp]
    lib.ut_fk.restype = i32
    lib.ut_fk.argtypes = [vp, f32p, i32, f32p, i32, f32p, i32, vp, ctypes.c_float, i32, f32p, vp]
    lib.ut_gen_crop_cameras.restype = i32
    lib.ut_gen_crop_cameras.argtypes = [vp, vp, vp, f32p, f32p, i32, f32p, f32p, vp, vp, i32, i32, i32, i32, i32, i32,
                                        i32, ctypes.c_double, vp, f32p, f32p, vp, vp, vp, f32p, vp]
    lib.ut_gen_crop_matrices.restype = i32
    lib.ut_gen_crop_matrices.argtypes = [vp, f32p, f32p, f32p, vp, i32, i32, i32, i32, ctypes.c_double, f32p, f32p, f32p,
                                         vp, vp]
    lib.ut_resample_homography.restype = i32
    lib.ut_resample_homography.argtypes = [vp, vp, i32, i32, i32, i32, f32p, i32, i32, f32p, vp]
    lib.ut_keypoint_metrics.restype = i32
    lib.ut_keypoint_metrics.argtypes = [vp, f32p, f32p, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.ut_profile_begin.restype = i32
    lib.ut_profile_begin.argtypes = [vp, vp]
    lib.ut_profile_end_by_kind.restype = i32
    lib.ut_profile_end_by_kind.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                           ctypes.POINTER(ctypes.c_double)]
    lib.ut_profile_end.restype = i32
    lib.ut_profile_end.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                   ctypes.POINTER(ctypes.c_double)]
    lib.ut_set_index_checks.restype = i32
    lib.ut_set_index_checks.argtypes = [vp, i32]
    lib.ut_set_backbone_lanes.restype = i32
    lib.ut_set_backbone_lanes.argtypes = [vp, i32]
    lib.ut_set_block_fusion.restype = i32
    lib.ut_set_block_fusion.argtypes = [vp, i32]
    lib.ut_set_resident_weights.restype = i32
    lib.ut_set_resident_weights.argtypes = [vp, i32]
    lib.ut_warp_map.restype = i32
    lib.ut_warp_map.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.ut_status_snapshot.restype = i32
    lib.ut_status_snapshot.argtypes = [vp, vp, vp]
    lib.ut_set_latency_mode.restype = i32
    lib.ut_set_latency_mode.argtypes = [vp, i32]
    lib.ut_set_conv_arithmetic.restype = i32
    lib.ut_set_conv_arithmetic.argtypes = [vp, i32]
    lib.ut_poll_status.restype = i32
    lib.ut_poll_status.argtypes = [vp, vp]
    lib.ut_set_split_scale.restype = i32
    lib.ut_set_split_scale.argtypes = [vp, i32]
    lib.ut_calibrate_split.restype = i32
    lib.ut_calibrate_split.argtypes = [vp, vp, i32, vp]
    lib.ut_get_split_calibration.restype = i32
    lib.ut_get_split_calibration.argtypes = [vp, vp]
    lib.ut_canonical_backbone_weights.restype = i32
    lib.ut_canonical_backbone_weights.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    _lib = lib
    return lib


def state_dict_to_blob(state_dict) -> np.ndarray:
    """Flatten a reference-keyed state dict (torch tensors or numpy arrays) in the order of
    arch.state_dict_spec() to the fp32 blob ut_create expects.  Strict like load_state_dict."""
    spec = arch.state_dict_spec()
    missing = [k for k, _s, _kind in spec if k not in state_dict]
    extra = [k for k in state_dict if k not in {k for k, _s, _kind in spec}]
    if missing or extra:
        raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:4]}..., "
                           f"unexpected keys {extra[:4]}..." if missing or extra else "")
    parts = []
    for k, shape, _kind in spec:
        v = state_dict[k]
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if tuple(v.shape) != tuple(shape):
            raise RuntimeError(f"size mismatch for {k}: expected {tuple(shape)}, got {tuple(v.shape)}")
        parts.append(v.astype(np.float32).reshape(-1))
    blob = np.ascontiguousarray(np.concatenate(parts))
    assert blob.size == 4_259_410
    return blob


def canonical_backbone_weights(state_dict) -> np.ndarray:
    """The backbone's folded convolutions at their canonical channel scales, as ut_create packs them (host only: runs
    without a GPU).  Layout: see ut_canonical_backbone_weights in include/umetrack_hip.h."""
    lib = load_library()
    blob = state_dict_to_blob(state_dict)
    n = ctypes.c_size_t()
    rc = lib.ut_canonical_backbone_weights(blob.ctypes.data_as(ctypes.c_void_p), blob.size, None, 0, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"ut_canonical_backbone_weights failed ({rc}): {lib.ut_last_error(None).decode()}")
    out = np.empty(n.value, np.float32)
    rc = lib.ut_canonical_backbone_weights(blob.ctypes.data_as(ctypes.c_void_p), blob.size,
                                           out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"ut_canonical_backbone_weights failed ({rc}): {lib.ut_last_error(None).decode()}")
    return out


def hand_model_blob(joint_rotation_axes, joint_rest_positions, landmark_rest_positions,
                    landmark_rest_bone_weights, landmark_rest_bone_indices) -> np.ndarray:
    """[...,321] fp32 packing of the HandModel fields the FK kernel reads (lib/common/hand.py:48-62)."""
    def a(x, tail):
        if isinstance(x, torch.Tensor):
            x = x.detach().cpu().numpy()
        x = np.asarray(x, np.float32)
        return x.reshape(x.shape[: x.ndim - len(tail)] + (-1,))
    parts = [a(joint_rotation_axes, (22, 3)), a(joint_rest_positions, (22, 3)), a(landmark_rest_positions, (21, 3)),
             a(landmark_rest_bone_weights, (21, 3)), a(landmark_rest_bone_indices, (21, 3))]
    lead = np.broadcast_shapes(*[p.shape[:-1] for p in parts])
    return np.ascontiguousarray(np.concatenate([np.broadcast_to(p, lead + p.shape[-1:]) for p in parts], -1))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need(t: torch.Tensor, dtype, device, name: str) -> torch.Tensor:
    if t.device != device:
        raise ValueError(f"{name} must live on {device}, got {t.device}")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def fk_stateless(hand_model: torch.Tensor, joint_angles: torch.Tensor, wrist_xf: torch.Tensor,
                 mirror: Optional[torch.Tensor] = None, t_scale: float = 1.0) -> torch.Tensor:
    """ut_fk without a model handle (the FK kernel needs no network weights).  All tensors on one HIP device."""
    lib = load_library()
    d = joint_angles.device
    if d.type != "cuda":
        raise NativeLibraryError("fk_stateless needs tensors on a HIP device (no CPU fallback)")
    hand_model = _need(hand_model, torch.float32, d, "hand_model").reshape(-1, 321)
    joint_angles = _need(joint_angles, torch.float32, d, "joint_angles").reshape(-1, 22)
    wrist_xf = _need(wrist_xf, torch.float32, d, "wrist_xf").reshape(-1, 16)
    n = joint_angles.shape[0]
    if mirror is not None:
        mirror = _need(mirror, torch.int64, d, "mirror").reshape(-1)
    out = torch.empty(n, arch.N_LANDMARKS, 3, dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        rc = lib.ut_fk(None, _ptr(hand_model), hand_model.shape[0], _ptr(joint_angles), 22, _ptr(wrist_xf), 16,
                       _ptr(mirror), ctypes.c_float(t_scale), n, _ptr(out), _stream(d))
    if rc != 0:
        raise RuntimeError(f"ut_fk failed ({rc}): {lib.ut_last_error(None).decode()}")
    return out


def gen_crop_cameras(cam_params: torch.Tensor, camera_angles: torch.Tensor, hand_model: torch.Tensor,
                     joint_limits: torch.Tensor, joint_angles: torch.Tensor, wrist_xf: torch.Tensor,
                     frame_idx: torch.Tensor, hand_idx: torch.Tensor, n_cams: int, src_wh: Tuple[int, int],
                     max_views: int = 2, min_vis: int = 19, crop_size: int = arch.CROP,
                     focal_multiplier: float = 0.8, check_indices: bool = True,
                     want_landmarks: bool = False) -> Dict[str, torch.Tensor]:
    """ut_gen_crop_cameras: crop cameras of n (frame, hand) label poses in one launch, padded to max_views.
    Returns crop_params [n,V,24] f64, intrinsics [n,V,3,3], extrinsics [n,V,4,4], cam_index [n,V] i32,
    n_views [n] i32, status [n] i32.  All tensors on one HIP device; no CPU fallback."""
    lib = load_library()
    d = joint_angles.device
    if d.type != "cuda":
        raise NativeLibraryError("gen_crop_cameras needs tensors on a HIP device (no CPU fallback)")
    cam_params = _need(cam_params, torch.float64, d, "cam_params").reshape(-1, 32)
    camera_angles = _need(camera_angles, torch.float64, d, "camera_angles").reshape(-1)
    hand_model = _need(hand_model, torch.float32, d, "hand_model").reshape(-1, 321)
    joint_limits = _need(joint_limits, torch.float32, d, "joint_limits").reshape(-1, 44)
    joint_angles = _need(joint_angles, torch.float32, d, "joint_angles").reshape(-1, 22)
    wrist_xf = _need(wrist_xf, torch.float32, d, "wrist_xf").reshape(-1, 16)
    frame_idx = _need(frame_idx, torch.int32, d, "frame_idx").reshape(-1)
    hand_idx = _need(hand_idx, torch.int64, d, "hand_idx").reshape(-1)
    n = joint_angles.shape[0]
    if not (wrist_xf.shape[0] == frame_idx.shape[0] == hand_idx.shape[0] == n):
        raise ValueError("gen_crop_cameras: joint_angles, wrist_xf, frame_idx and hand_idx disagree on n")
    if hand_model.shape[0] != joint_limits.shape[0] or hand_model.shape[0] not in (1, n):
        raise ValueError("gen_crop_cameras: hand_model / joint_limits must hold 1 or n models")
    if camera_angles.shape[0] != n_cams or cam_params.shape[0] % n_cams:
        raise ValueError("gen_crop_cameras: cam_params rows must be a multiple of n_cams = len(camera_angles)")
    # (reads frame_idx back: pass check_indices=False when the same index tensor was validated before)
    if check_indices and n and (int(frame_idx.max()) + 1) * n_cams > cam_params.shape[0]:
        raise ValueError("gen_crop_cameras: frame_idx points past cam_params")
    out = {"crop_params": torch.zeros(n, max_views, 24, dtype=torch.float64, device=d),
           "intrinsics": torch.zeros(n, max_views, 3, 3, dtype=torch.float32, device=d),
           "extrinsics": torch.zeros(n, max_views, 4, 4, dtype=torch.float32, device=d),
           "cam_index": torch.empty(n, max_views, dtype=torch.int32, device=d),
           "n_views": torch.empty(n, dtype=torch.int32, device=d),
           "status": torch.empty(n, dtype=torch.int32, device=d)}
    if want_landmarks:
        out["landmarks"] = torch.empty(n, arch.N_LANDMARKS, 3, dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        rc = lib.ut_gen_crop_cameras(None, _ptr(cam_params), _ptr(camera_angles), _ptr(hand_model), _ptr(joint_limits),
                                     hand_model.shape[0], _ptr(joint_angles), _ptr(wrist_xf), _ptr(frame_idx),
                                     _ptr(hand_idx), n, n_cams, max_views, min_vis, int(src_wh[0]), int(src_wh[1]),
                                     crop_size, ctypes.c_double(focal_multiplier), _ptr(out["crop_params"]),
                                     _ptr(out["intrinsics"]), _ptr(out["extrinsics"]), _ptr(out["cam_index"]),
                                     _ptr(out["n_views"]), _ptr(out["status"]), _ptr(out.get("landmarks")), _stream(d))
    if rc != 0:
        raise RuntimeError(f"ut_gen_crop_cameras failed ({rc}): {lib.ut_last_error(None).decode()}")
    return out


def gen_crop_matrices(orig_extrinsics: torch.Tensor, orig_intrinsics: torch.Tensor, crop_points: torch.Tensor,
                      hand_idx: torch.Tensor, crop_size: int = arch.CROP, focal_multiplier: float = 0.95
                      ) -> Dict[str, torch.Tensor]:
    """ut_gen_crop_matrices.  orig_extrinsics [F,V,4,4], orig_intrinsics [F,V,3,3], crop_points [F,P,3], hand_idx [F]
    -> extrinsics_xf [F,V,4,4], new_intrinsics [F,V,3,3], resample_xf [F,V,4,4], status [F,V] (i32)."""
    lib = load_library()
    d = orig_extrinsics.device
    if d.type != "cuda":
        raise NativeLibraryError("gen_crop_matrices needs tensors on a HIP device (no CPU fallback)")
    if orig_extrinsics.dim() != 4 or tuple(orig_extrinsics.shape[2:]) != (4, 4):
        raise ValueError("orig_extrinsics must be [frames, views, 4, 4]")
    f, v = orig_extrinsics.shape[:2]
    if tuple(orig_intrinsics.shape) != (f, v, 3, 3):
        raise ValueError("orig_intrinsics must be [frames, views, 3, 3]")
    if crop_points.dim() != 3 or crop_points.shape[0] != f or crop_points.shape[2] != 3 or crop_points.shape[1] < 1:
        raise ValueError("crop_points must be [frames, points, 3]")
    ext = _need(orig_extrinsics, torch.float32, d, "orig_extrinsics")
    intr = _need(orig_intrinsics, torch.float32, d, "orig_intrinsics")
    pts = _need(crop_points, torch.float32, d, "crop_points")
    hand = _need(hand_idx, torch.int64, d, "hand_idx").reshape(-1)
    if hand.shape[0] != f:
        raise ValueError("hand_idx must be [frames]")
    out = {"extrinsics_xf": torch.empty(f, v, 4, 4, device=d), "new_intrinsics": torch.empty(f, v, 3, 3, device=d),
           "resample_xf": torch.empty(f, v, 4, 4, device=d), "status": torch.empty(f, v, dtype=torch.int32, device=d)}
    with torch.cuda.device(d):
        rc = lib.ut_gen_crop_matrices(None, _ptr(ext), _ptr(intr), _ptr(pts), _ptr(hand), f, v, pts.shape[1], crop_size,
                                      ctypes.c_double(focal_multiplier), _ptr(out["extrinsics_xf"]),
                                      _ptr(out["new_intrinsics"]), _ptr(out["resample_xf"]), _ptr(out["status"]), _stream(d))
    if rc != 0:
        raise RuntimeError(f"ut_gen_crop_matrices failed ({rc}): {lib.ut_last_error(None).decode()}")
    return out


def resample_homography(src: torch.Tensor, resample_xf: torch.Tensor, out_hw: Tuple[int, int] = (arch.CROP, arch.CROP),
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ut_resample_homography.  src [n,H,W] u8 or f32, resample_xf [n,4,4] f32 -> [n,h,w] f32 in [0,1]."""
    lib = load_library()
    d = src.device
    if d.type != "cuda":
        raise NativeLibraryError("resample_homography needs tensors on a HIP device (no CPU fallback)")
    if src.dim() != 3 or src.dtype not in (torch.uint8, torch.float32):
        raise ValueError("src must be [n,H,W] uint8 or float32")
    n = src.shape[0]
    xf = _need(resample_xf, torch.float32, d, "resample_xf").reshape(-1, 16)
    if xf.shape[0] != n:
        raise ValueError("resample_xf must hold one 4x4 matrix per source image")
    src = src.contiguous()
    if out is None:
        out = torch.empty(n, out_hw[0], out_hw[1], device=d)
    elif tuple(out.shape) != (n, out_hw[0], out_hw[1]) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != d:
        raise ValueError("out must be a contiguous float32 [n,h,w] tensor on the source's device")
    with torch.cuda.device(d):
        rc = lib.ut_resample_homography(None, _ptr(src), int(src.dtype == torch.float32), n, src.shape[1], src.shape[2],
                                        _ptr(xf), out_hw[0], out_hw[1], _ptr(out), _stream(d))
    if rc != 0:
        raise RuntimeError(f"ut_resample_homography failed ({rc}): {lib.ut_last_error(None).decode()}")
    return out


def keypoint_metrics(gt: torch.Tensor, tracked: torch.Tensor, valid: torch.Tensor) -> Dict[str, torch.Tensor]:
    """ut_keypoint_metrics.  gt, tracked [hands, frames, 21, 3] f32, valid [hands, frames] bool/u8 on one HIP device ->
    err f64 [hands, frames], acc / gt_acc f64 [hands, frames-2], valid_acc bool [hands, frames-2]."""
    lib = load_library()
    d = gt.device
    if d.type != "cuda":
        raise NativeLibraryError("keypoint_metrics needs tensors on a HIP device (no CPU fallback)")
    if gt.dim() != 4 or tuple(gt.shape[2:]) != (arch.N_LANDMARKS, 3) or gt.shape != tracked.shape:
        raise ValueError("gt and tracked must both be [hands, frames, 21, 3]")
    h, t = gt.shape[:2]
    if tuple(valid.shape) != (h, t):
        raise ValueError("valid must be [hands, frames]")
    g, p = _need(gt, torch.float32, d, "gt"), _need(tracked, torch.float32, d, "tracked")
    v = _need(valid, torch.uint8, d, "valid")
    ta = max(t - 2, 0)
    out = {"err": torch.empty(h, t, dtype=torch.float64, device=d), "acc": torch.empty(h, ta, dtype=torch.float64, device=d),
           "gt_acc": torch.empty(h, ta, dtype=torch.float64, device=d), "valid_acc": torch.zeros(h, ta, dtype=torch.uint8, device=d)}
    with torch.cuda.device(d):
        rc = lib.ut_keypoint_metrics(None, _ptr(g), _ptr(p), _ptr(v), h, t, _ptr(out["err"]), _ptr(out["acc"]),
                                     _ptr(out["gt_acc"]), _ptr(out["valid_acc"]), _stream(d))
    if rc != 0:
        raise RuntimeError(f"ut_keypoint_metrics failed ({rc}): {lib.ut_last_error(None).decode()}")
    out["valid_acc"] = out["valid_acc"].bool()
    return out


def warp_map(cam_params: torch.Tensor, crop_params: torch.Tensor, src_index: torch.Tensor, n_src_images: int) -> torch.Tensor:
    """ut_warp_map: the fp32 coordinate maps [n_crops,96,96,2] (x, y) the resampler samples with (what the reference hands to
    cv2.remap, lib/tracker/tracker.py:69-85)."""
    lib = load_library()
    d = cam_params.device
    if d.type != "cuda":
        raise NativeLibraryError("warp_map needs tensors on a HIP device (no CPU fallback)")
    cam, crop = _need(cam_params, torch.float64, d, "cam_params"), _need(crop_params, torch.float64, d, "crop_params")
    idx = _need(src_index, torch.int32, d, "src_index")
    n = crop.shape[0]
    if crop.shape != (n, 24) or cam.dim() != 2 or cam.shape[1] != 32 or idx.shape != (n,) or cam.shape[0] < n_src_images:
        raise ValueError("cam_params [n_src,32] f64, crop_params [n,24] f64, src_index [n] i32")
    out = torch.empty(n, arch.CROP, arch.CROP, 2, device=d)
    with torch.cuda.device(d):
        rc = lib.ut_warp_map(_ptr(cam), _ptr(crop), _ptr(idx), int(n_src_images), n, _ptr(out), _stream(d))
    if rc != 0:
        raise RuntimeError(f"ut_warp_map failed ({rc}): {lib.ut_last_error(None).decode()}")
    return out


class HipEngine:
    """One native handle (packed weights + workspace + temporal state) on one GPU."""

    def __init__(self, state_dict, device="cuda"):
        if not torch.cuda.is_available():
            raise NativeLibraryError("no HIP device visible: the UmeTrack hot path has no CPU fallback")
        self.lib = load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise NativeLibraryError(f"device {device!r} is not a HIP device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        blob = state_dict_to_blob(state_dict)
        h = ctypes.c_void_p()
        rc = self.lib.ut_create(self.device.index, blob.ctypes.data_as(ctypes.c_void_p), blob.size, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"ut_create failed ({rc}): {self.lib.ut_last_error(None).decode()}")
        self._h = h
        self.deferred_checks = False      # mirrors of the handle's modes (the C ABI has setters only)
        self.latency_mode = False

    def close(self):
        if getattr(self, "_h", None):
            self.lib.ut_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc < 0:
            msg = self.lib.ut_last_error(self._h).decode()
            if rc == -4:
                raise AssertionError(msg)      # the reference asserts here (umetrack_model.py:224-229)
            if rc == -1 and "index check:" in msg:
                raise IndexError(msg)          # the reference's tensor indexing raises IndexError on these
            if rc == -1 and "range check:" in msg:
                raise FloatingPointError(msg)  # split-fp16 backbone: an infinity / a NaN among a layer's input activations
            raise RuntimeError(f"{what} failed ({rc}): {msg}")
        return rc

    def set_index_checks(self, deferred: bool):
        """Default: every call that takes index tensors reads the device-side verdict back (one stream sync) and
        raises IndexError.  deferred=True: no sync (hipGraph capture, run-ahead launching); bad work is skipped on
        the device and `poll_status()` raises for it later."""
        self._check(self.lib.ut_set_index_checks(self._h, UT_CHECK_DEFERRED if deferred else UT_CHECK_SYNC),
                    "ut_set_index_checks")
        self.deferred_checks = bool(deferred)

    @contextlib.contextmanager
    def modes(self, deferred_checks: Optional[bool] = None, latency: Optional[bool] = None):
        """Switch the handle's check / latency mode for the calls inside the `with` block and put back what was set
        before: a handle shared by a per-frame HandTracker (latency mode, deferred checks) and a batched HotPath keeps
        each user's settings out of the other's calls."""
        prev = (self.deferred_checks, self.latency_mode)
        try:
            if deferred_checks is not None and deferred_checks != prev[0]:
                self.set_index_checks(deferred_checks)
            if latency is not None and latency != prev[1]:
                self.set_latency_mode(latency)
            yield self
        finally:
            if self._h:
                if self.deferred_checks != prev[0]:
                    self.set_index_checks(prev[0])
                if self.latency_mode != prev[1]:
                    self.set_latency_mode(prev[1])

    def status_snapshot(self, out: torch.Tensor):
        """Stream-ordered copy of the two device status words (sticky errors, this call's bits) into `out` (int32 [2] on
        the device) - for callers that read results back in one staged transfer and look at the verdict there."""
        out = _need(out, torch.int32, self.device, "out")
        self._check(self.lib.ut_status_snapshot(self._h, _ptr(out), _stream(self.device)), "ut_status_snapshot")

    def set_backbone_lanes(self, lanes: int):
        """2: large batches run as two half-batches on two internal streams (each fills the other's launch tails)."""
        self._check(self.lib.ut_set_backbone_lanes(self._h, int(lanes)), "ut_set_backbone_lanes")

    def set_block_fusion(self, on: bool):
        """Split-fp16 mode: layer1's BasicBlocks as one launch each (default) or as two convolution launches (A/B tests)."""
        self._check(self.lib.ut_set_block_fusion(self._h, int(bool(on))), "ut_set_block_fusion")

    def set_resident_weights(self, kind=1):
        """Split-fp16 mode, A/B switch (include/umetrack_hip.h::ut_set_resident_weights): 1 / True (default) conv_w4.hip on the
        stride-1 convolutions of layer2 .. layer4 and the stride-2 entries of layer3 / layer4; 0 / False the chunked conv_split
        kernels everywhere; 6 as 1 but the stride-2 entries through the chunked gather kernel; and, with those through the gather
        kernel as well: 3 conv_w4 on all stride-1 convolutions, 2 conv_c64k.hip on layer2 and chunked elsewhere, 4 conv_w4 on
        layer3 / layer4 and conv_c64k on layer2, 5 conv_w4 on layer3 / layer4 and chunked on layer2."""
        self._check(self.lib.ut_set_resident_weights(self._h, int(kind)), "ut_set_resident_weights")

    def set_latency_mode(self, on: bool):
        """Few-crop launches split K across workgroups (per-frame tracking); results then agree with the default mode to
        fp32 rounding instead of bit for bit.  Off by default."""
        self._check(self.lib.ut_set_latency_mode(self._h, int(bool(on))), "ut_set_latency_mode")
        self.latency_mode = bool(on)

    def set_conv_arithmetic(self, mode: str):
        """"fp32": exact fp32 matrix instructions (default).  "split_f16": the batched backbone convolutions run on the fp16
        matrix cores from two-piece splits of both operands (fp32-level error, not the fp32 chain's bits), for calls of
        >= 2 x CUs crops (one arithmetic per call); "split_f16_always": calls of any size (tests)."""
        self._check(self.lib.ut_set_conv_arithmetic(self._h, {"fp32": 0, "split_f16": 1, "split_f16_always": 2}[mode]), "ut_set_conv_arithmetic")

    def set_split_scale(self, mode: str):
        """Split-fp16 mode: "calibrated" (default) - one fixed power-of-two activation scale per backbone tensor and handle (a
        crop's bits do not depend on its batch, lane count or sharding; inputs beyond 32 x the calibration maximum are reported
        by poll_status as FloatingPointError) - or "dynamic": each launch scales by the largest magnitude its producer stored in
        this call."""
        self._check(self.lib.ut_set_split_scale(self._h, {"calibrated": 0, "dynamic": 1}[mode]), "ut_set_split_scale")

    def calibrate_split(self, crops: Optional[torch.Tensor] = None):
        """Take the calibrated activation scales from `crops` ([n,96,96] fp32 on the device; None: the built-in synthetic set,
        which is what a handle uses when this is never called)."""
        if crops is None:
            self._check(self.lib.ut_calibrate_split(self._h, None, 0, _stream(self.device)), "ut_calibrate_split")
            return
        crops = _need(crops, torch.float32, self.device, "crops")
        if crops.ndim != 3 or tuple(crops.shape[1:]) != (arch.CROP, arch.CROP) or crops.shape[0] == 0:
            raise ValueError(f"crops must be [n >= 1,{arch.CROP},{arch.CROP}], got {tuple(crops.shape)}")
        self._check(self.lib.ut_calibrate_split(self._h, _ptr(crops), crops.shape[0], _stream(self.device)), "ut_calibrate_split")

    def split_calibration(self) -> np.ndarray:
        """The 33 calibrated scale words (stem output, every block's inner tensor and output, 2 x 4 of the regressors) as float32."""
        out = np.zeros(33, np.float32)
        self._check(self.lib.ut_get_split_calibration(self._h, out.ctypes.data_as(ctypes.c_void_p)), "ut_get_split_calibration")
        return out

    def poll_status(self):
        self._check(self.lib.ut_poll_status(self._h, _stream(self.device)), "ut_poll_status")

    # ------------------------------------------------------------------ entry points
    def reserve(self, max_crops: int, max_samples: int, max_slots: int):
        self._check(self.lib.ut_reserve(self._h, max_crops, max_samples, max_slots), "ut_reserve")

    def set_backbone_chunk(self, crops: int):
        self._check(self.lib.ut_set_backbone_chunk(self._h, crops), "ut_set_backbone_chunk")

    def warp_crops(self, src_u8: torch.Tensor, cam_params: torch.Tensor, crop_params: torch.Tensor,
                   src_index: torch.Tensor, mode: int = UT_REMAP_CV2_FIXED,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        d = self.device
        src_u8 = _need(src_u8, torch.uint8, d, "src")
        if src_u8.dim() != 3:
            raise ValueError("src must be [n_images, H, W] uint8")
        cam_params = _need(cam_params, torch.float64, d, "cam_params")
        crop_params = _need(crop_params, torch.float64, d, "crop_params")
        src_index = _need(src_index, torch.int32, d, "src_index")
        n = crop_params.shape[0]
        if cam_params.shape != (src_u8.shape[0], 32) or crop_params.shape != (n, 24) or src_index.shape != (n,):
            raise ValueError("bad cam_params / crop_params / src_index shape")
        if out is None:
            out = torch.empty(n, arch.CROP, arch.CROP, dtype=torch.float32, device=d)
        self._check(self.lib.ut_warp_crops(self._h, _ptr(src_u8), src_u8.shape[0], src_u8.shape[1], src_u8.shape[2],
                                           _ptr(cam_params), _ptr(crop_params), _ptr(src_index), n, mode, _ptr(out),
                                           _stream(d)), "ut_warp_crops")
        return out

    def warp_backbone(self, src_u8: torch.Tensor, cam_params: torch.Tensor, crop_params: torch.Tensor,
                      src_index: torch.Tensor, mode: int = UT_REMAP_CV2_FIXED,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ut_warp_backbone: resample + backbone with the crops kept in the handle's workspace (u8 in cv2 mode)."""
        d = self.device
        src_u8 = _need(src_u8, torch.uint8, d, "src")
        if src_u8.dim() != 3:
            raise ValueError("src must be [n_images, H, W] uint8")
        cam_params = _need(cam_params, torch.float64, d, "cam_params")
        crop_params = _need(crop_params, torch.float64, d, "crop_params")
        src_index = _need(src_index, torch.int32, d, "src_index")
        n = crop_params.shape[0]
        if cam_params.shape != (src_u8.shape[0], 32) or crop_params.shape != (n, 24) or src_index.shape != (n,):
            raise ValueError("bad cam_params / crop_params / src_index shape")
        if out is None:
            out = torch.empty(n, arch.FEAT_CH, arch.FEAT_HW, arch.FEAT_HW, dtype=torch.float32, device=d)
        self._check(self.lib.ut_warp_backbone(self._h, _ptr(src_u8), src_u8.shape[0], src_u8.shape[1], src_u8.shape[2],
                                              _ptr(cam_params), _ptr(crop_params), _ptr(src_index), n, mode, _ptr(out),
                                              _stream(d)), "ut_warp_backbone")
        return out

    def backbone(self, crops: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        d = self.device
        crops = _need(crops, torch.float32, d, "crops")
        if crops.dim() != 3 or crops.shape[1:] != (arch.CROP, arch.CROP):
            raise ValueError(f"crops must be [n,96,96], got {tuple(crops.shape)}")
        n = crops.shape[0]
        if out is None:
            out = torch.empty(n, arch.FEAT_CH, arch.FEAT_HW, arch.FEAT_HW, dtype=torch.float32, device=d)
        self._check(self.lib.ut_backbone(self._h, _ptr(crops), n, _ptr(out), _stream(d)), "ut_backbone")
        return out

    def fuse_temporal_regress(self, feat, intrinsics, extrinsics, sample_range, memory_idx, use_memory, hand_idx,
                              n_slots: int, all_multiview: bool, skel: Optional[torch.Tensor], mode: int,
                              want_raw: bool = False, out: Optional[torch.Tensor] = None
                              ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        d = self.device
        feat = _need(feat, torch.float32, d, "feat")
        intrinsics = _need(intrinsics, torch.float32, d, "intrinsics")
        extrinsics = _need(extrinsics, torch.float32, d, "extrinsics")
        sample_range = _need(sample_range, torch.int64, d, "sample_range")
        memory_idx = _need(memory_idx, torch.int64, d, "memory_idx")
        hand_idx = _need(hand_idx, torch.int64, d, "hand_idx")
        use_memory = _need(use_memory.to(torch.uint8) if use_memory.dtype == torch.bool else use_memory,
                           torch.uint8, d, "use_memory")
        n, s = feat.shape[0], sample_range.shape[0]
        if intrinsics.shape != (n, 3, 3) or extrinsics.shape != (n, 4, 4) or sample_range.shape != (s, 2) \
                or memory_idx.shape != (s,) or use_memory.shape != (s,) or hand_idx.shape != (s,):
            raise ValueError("inconsistent frame data / frame desc shapes")
        n_skel = 0
        if skel is not None:
            skel = _need(skel, torch.float32, d, "skel")
            n_skel = skel.shape[0]
        if out is None:
            out = torch.empty(s, arch.POSE_REC, dtype=torch.float32, device=d)
        raw = torch.empty(s, 64, dtype=torch.float32, device=d) if want_raw else None
        self._check(self.lib.ut_fuse_temporal_regress(
            self._h, _ptr(feat), _ptr(intrinsics), _ptr(extrinsics), _ptr(sample_range), _ptr(memory_idx),
            _ptr(use_memory), _ptr(hand_idx), n, s, int(n_slots), int(bool(all_multiview)), _ptr(skel), n_skel, mode,
            _ptr(out), _ptr(raw), _stream(d)), "ut_fuse_temporal_regress")
        return out, raw

    def reset_memory(self):
        self._check(self.lib.ut_reset_memory(self._h), "ut_reset_memory")

    def get_memory(self, max_slots: int = 1 << 16):
        d = self.device
        n = self._check(self.lib.ut_get_memory(self._h, None, None, 0, _stream(d)), "ut_get_memory")
        n = min(n, max_slots)
        mem = torch.empty(n, arch.MEM_CH, arch.FEAT_HW, arch.FEAT_HW, dtype=torch.float32, device=d)
        ext = torch.empty(n, 4, 4, dtype=torch.float32, device=d)
        if n:
            self._check(self.lib.ut_get_memory(self._h, _ptr(mem), _ptr(ext), n, _stream(d)), "ut_get_memory")
        return mem, ext

    def fk(self, hand_model: torch.Tensor, joint_angles: torch.Tensor, wrist_xf: torch.Tensor,
           mirror: Optional[torch.Tensor] = None, t_scale: float = 1.0, ja_stride: int = 22, xf_stride: int = 16,
           n: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """hand_model [1|n,321]; joint_angles/wrist_xf either packed [n,22]/[n,4,4] or views into a pose
        record buffer with explicit strides (in floats)."""
        d = self.device
        hand_model = _need(hand_model, torch.float32, d, "hand_model").reshape(-1, 321)
        if n is None:
            joint_angles = _need(joint_angles, torch.float32, d, "joint_angles").reshape(-1, 22)
            wrist_xf = _need(wrist_xf, torch.float32, d, "wrist_xf").reshape(-1, 16)
            n = joint_angles.shape[0]
            if wrist_xf.shape[0] != n:
                raise ValueError("joint_angles / wrist_xf batch mismatch")
        if mirror is not None:
            mirror = _need(mirror, torch.int64, d, "mirror").reshape(-1)
            if mirror.shape[0] != n:
                raise ValueError("mirror batch mismatch")
        if out is None:
            out = torch.empty(n, arch.N_LANDMARKS, 3, dtype=torch.float32, device=d)
        self._check(self.lib.ut_fk(self._h, _ptr(hand_model), hand_model.shape[0], _ptr(joint_angles), ja_stride,
                                   _ptr(wrist_xf), xf_stride, _ptr(mirror), ctypes.c_float(t_scale), n, _ptr(out),
                                   _stream(d)), "ut_fk")
        return out

    def profile_begin(self):
        self._check(self.lib.ut_profile_begin(self._h, _stream(self.device)), "ut_profile_begin")

    def profile_end_by_kind(self):
        """[(ms, launches, flops) of the fp32-matrix-instruction launches, (...) of the split-fp16 launches]"""
        ms, n, fl = (ctypes.c_double * 2)(), (ctypes.c_int64 * 2)(), (ctypes.c_double * 2)()
        self._check(self.lib.ut_profile_end_by_kind(self._h, _stream(self.device), ms, n, fl), "ut_profile_end_by_kind")
        return [(ms[k], n[k], fl[k]) for k in range(2)]

    def profile_end(self):
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        self._check(self.lib.ut_profile_end(self._h, _stream(self.device), ctypes.byref(ms), ctypes.byref(n),
                                            ctypes.byref(fl)), "ut_profile_end")
        return ms.value, n.value, fl.value
