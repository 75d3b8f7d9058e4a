"""Per-frame tracker with the reference's call surface (lib/tracker/tracker.py:40-412,
lib/tracker/perspective_crop.py:19-180, lib/tracker/tracking_result.py:14-30).

Host code here is parameter plumbing only: crop-camera parameters (a few 4x4 products per hand),
packing of kernel argument rows, dict bookkeeping.  The heavy steps run natively:
  * forward kinematics of the crop points / landmarks  -> csrc/fk.hip    (ut_fk)
  * fisheye->pinhole resampling of every crop          -> csrc/warp.hip  (ut_warp_crops)
  * the network                                        -> ut_backbone + ut_fuse_temporal_regress
"""
import logging
from dataclasses import dataclass
from typing import Dict, List, NamedTuple, Optional, Tuple

import numpy as np
import torch

from . import _native, geometry
from .geometry import CameraModel, PinholePlaneCameraModel
from .hand import NUM_HANDS, NUM_JOINTS_PER_HAND, RIGHT_HAND_INDEX, HandModel, scaled_hand_model, skin_landmarks
from .model import InputFrameData, InputFrameDesc, InputSkeletonData, RegressorOutput

logger = logging.getLogger(__name__)

MM_TO_M = 0.001
M_TO_MM = 1000.0
MIN_OBSERVED_LANDMARKS = 21
CONFIDENCE_THRESHOLD = 0.5
MAX_VIEW_NUM = 2


# ----------------------------------------------------------------------------- tracking_result.py
class SingleHandPose(NamedTuple):
    """joint angles (one per DoF) + root-to-world wrist transform (mm)."""
    joint_angles: np.ndarray = np.zeros(NUM_JOINTS_PER_HAND, dtype=np.float32)
    wrist_xform: np.ndarray = np.eye(4, dtype=np.float32)
    hand_confidence: float = 1.0


class TrackingResult(NamedTuple):
    hand_poses: Dict[int, SingleHandPose] = {}
    num_views: Dict[int, int] = {}
    predicted_scales: Dict[int, float] = {}


# ----------------------------------------------------------------------------- perspective_crop.py
def neutral_joint_angles(up: HandModel, lower_factor: float = 0.5) -> torch.Tensor:
    lim = up.joint_limits
    assert lim is not None
    return lim[..., 0] * lower_factor + lim[..., 1] * (1 - lower_factor)


def skin_landmarks_np(hand_model: HandModel, joint_angles: np.ndarray, wrist_transforms: np.ndarray) -> np.ndarray:
    out = skin_landmarks(hand_model, torch.from_numpy(np.asarray(joint_angles)).float(),
                         torch.from_numpy(np.asarray(wrist_transforms)).float())
    return out.cpu().numpy()


def _left_handed(wrist_xform: np.ndarray, hand_idx: int) -> np.ndarray:
    xf = np.array(wrist_xform, copy=True)
    if hand_idx == RIGHT_HAND_INDEX:      # the hand model is a left hand: mirror x for right hands
        xf[:, 0] *= -1
    return xf


def landmarks_from_hand_pose(hand_model: HandModel, hand_pose: SingleHandPose, hand_idx: int) -> np.ndarray:
    """World-space landmarks [21,3] of a pose (lib/tracker/perspective_crop.py:40-51)."""
    return skin_landmarks_np(hand_model, hand_pose.joint_angles, _left_handed(hand_pose.wrist_xform, hand_idx))


def _visible_counts(cameras: List[CameraModel], landmarks_world: np.ndarray) -> List[int]:
    counts = []
    for cam in cameras:
        eye = cam.world_to_eye(landmarks_world)
        win = cam.eye_to_window(eye)
        inside = ((win[..., 0] >= 0) & (win[..., 0] <= cam.width - 1) & (win[..., 1] >= 0)
                  & (win[..., 1] <= cam.height - 1) & (eye[..., 2] > 0))
        counts.append(int(inside.sum()))
    return counts


def rank_hand_visibility_in_cameras(cameras, hand_model, hand_pose, hand_idx, min_required_vis_landmarks) -> List[int]:
    counts = _visible_counts(cameras, landmarks_from_hand_pose(hand_model, hand_pose, hand_idx))
    keep = [i for i, n in enumerate(counts) if n >= min_required_vis_landmarks]
    keep.sort(reverse=True, key=lambda i: counts[i])
    return keep


def _pose_stack_landmarks(hand_model: HandModel, poses: List[np.ndarray], wrist_xform: np.ndarray, hand_idx: int
                          ) -> np.ndarray:
    """FK of several joint-angle vectors under one wrist transform in ONE kernel launch: [len(poses),21,3]."""
    ja = np.stack([np.asarray(p, np.float32) for p in poses])
    xf = np.broadcast_to(_left_handed(wrist_xform, hand_idx).astype(np.float32), (len(poses), 4, 4))
    # an unbatched model is shared by all poses of the launch (ut_fk n_models == 1)
    return skin_landmarks_np(hand_model, ja, np.ascontiguousarray(xf))


def _get_crop_points_from_hand_pose(hand_model, gt_hand_pose, hand_idx, num_crop_points) -> np.ndarray:
    assert num_crop_points in [21, 42, 63]
    poses = [gt_hand_pose.joint_angles]
    if num_crop_points > 21:
        poses.append(neutral_joint_angles(hand_model).numpy())
    if num_crop_points > 42:
        poses.append(np.zeros(NUM_JOINTS_PER_HAND, dtype=np.float32))
    return _pose_stack_landmarks(hand_model, poses, gt_hand_pose.wrist_xform, hand_idx).reshape(-1, 3)


def gen_crop_cameras_from_pose(cameras, camera_angles, hand_model, hand_pose, hand_idx, num_crop_points,
                               new_image_size, max_view_num: Optional[int] = None, sort_camera_index: bool = False,
                               focal_multiplier: float = 0.95, mirror_right_hand: bool = True,
                               min_required_vis_landmarks: int = 19) -> Dict[int, PinholePlaneCameraModel]:
    """Pick the best views of one hand and aim a 96x96 pinhole crop camera at it from each
    (lib/tracker/perspective_crop.py:136-180).  The first 21 crop points are the landmarks of the pose
    itself, so the visibility ranking re-uses them instead of running FK again."""
    crop_points = _get_crop_points_from_hand_pose(hand_model, hand_pose, hand_idx, num_crop_points)
    counts = _visible_counts(cameras, crop_points[:21])
    order = [i for i, n in enumerate(counts) if n >= min_required_vis_landmarks]
    order.sort(reverse=True, key=lambda i: counts[i])
    if sort_camera_index:
        order = sorted(order)
    out: Dict[int, PinholePlaneCameraModel] = {}
    for ci in order:
        out[ci] = geometry.gen_crop_parameters_from_points(
            cameras[ci], crop_points, new_image_size, mirror_img_x=(mirror_right_hand and hand_idx == 1),
            camera_angle=camera_angles[ci], focal_multiplier=focal_multiplier)
        if len(out) == max_view_num:
            break
    return out


# ----------------------------------------------------------------------------- tracker.py
@dataclass
class ViewData:
    image: np.ndarray
    camera: CameraModel
    camera_angle: float


@dataclass
class InputFrame:
    views: List[ViewData]


@dataclass
class HandTrackerOpts:
    num_crop_points: int = 63
    enable_memory: bool = True
    use_stored_pose_for_crop: bool = True
    hand_ratio_in_crop: float = 0.8
    min_required_vis_landmarks: int = 19


def network_camera_inputs(crop_camera: PinholePlaneCameraModel) -> Tuple[np.ndarray, np.ndarray]:
    """K [3,3] and world->eye extrinsics with the translation in metres (lib/tracker/tracker.py:333-337)."""
    ext = np.linalg.inv(crop_camera.camera_to_world_xf)
    ext[:3, 3] *= MM_TO_M
    return crop_camera.uv_to_window_matrix(), ext


class HandTracker:
    def __init__(self, model, opts: HandTrackerOpts) -> None:
        self._device: str = "cuda" if torch.cuda.device_count() else "cpu"
        logger.info(f"Using device: {self._device}")
        self._model = model
        self._model.to(self._device)
        self._input_size = np.array(self._model.getInputImageSizes())
        self._num_crop_points = opts.num_crop_points
        self._enable_memory = opts.enable_memory
        self._hand_ratio_in_crop: float = opts.hand_ratio_in_crop
        self._min_required_vis_landmarks: int = opts.min_required_vis_landmarks
        self._valid_tracking_history = np.zeros(2, dtype=bool)
        self._remap_mode = _native.UT_REMAP_CV2_FIXED

    def reset_history(self) -> None:
        self._valid_tracking_history[:] = False

    def gen_crop_cameras(self, cameras: List[CameraModel], camera_angles: List[float], hand_model: HandModel,
                         gt_tracking: Dict[int, SingleHandPose], min_num_crops: int
                         ) -> Dict[int, Dict[int, PinholePlaneCameraModel]]:
        crop_cameras: Dict[int, Dict[int, PinholePlaneCameraModel]] = {}
        hands = [(h, p) for h, p in (gt_tracking or {}).items() if p.hand_confidence >= CONFIDENCE_THRESHOLD]
        if hands and self._device == "cuda" and self._batched_cropgen_ok(cameras, hand_model):
            return self._gen_crop_cameras_batched(cameras, camera_angles, hand_model, hands, min_num_crops)
        for hand_idx, pose in (gt_tracking or {}).items():
            if pose.hand_confidence < CONFIDENCE_THRESHOLD:
                continue
            per_hand = gen_crop_cameras_from_pose(
                cameras, camera_angles, hand_model, pose, hand_idx, self._num_crop_points, self._input_size,
                max_view_num=MAX_VIEW_NUM, sort_camera_index=True, focal_multiplier=self._hand_ratio_in_crop,
                mirror_right_hand=True, min_required_vis_landmarks=self._min_required_vis_landmarks)
            if per_hand and len(per_hand) >= min_num_crops:
                crop_cameras[hand_idx] = per_hand
        return crop_cameras

    def _batched_cropgen_ok(self, cameras, hand_model) -> bool:
        """ut_gen_crop_cameras covers the configuration the eval scripts use: 63 crop points, square crops, Fisheye62
        source cameras of one size, an unbatched hand model with joint limits."""
        return (self._num_crop_points == 63 and self._input_size[0] == self._input_size[1] and len(cameras) > 0
                and all(isinstance(c, geometry.Fisheye62CameraModel) for c in cameras)
                and len({(c.width, c.height) for c in cameras}) == 1
                and hand_model.joint_limits is not None and hand_model.joint_rest_positions.dim() == 2)

    def _gen_crop_cameras_batched(self, cameras, camera_angles, hand_model, hands, min_num_crops):
        """All hands of the frame through one ut_gen_crop_cameras launch (lib/tracker/tracker.py:222-260)."""
        from .hand import device_blob
        dev = torch.device("cuda", torch.cuda.current_device())
        n = len(hands)
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        cam_rows = np.stack([geometry.pack_camera_model(c) for c in cameras])
        g = _native.gen_crop_cameras(
            up(cam_rows, np.float64), up(np.asarray(camera_angles), np.float64), device_blob(hand_model, dev),
            hand_model.joint_limits.float().to(dev),
            up(np.stack([np.asarray(p.joint_angles) for _h, p in hands]), np.float32),
            up(np.stack([np.asarray(p.wrist_xform) for _h, p in hands]), np.float32),
            torch.zeros(n, dtype=torch.int32, device=dev), up(np.array([h for h, _p in hands]), np.int64),
            len(cameras), (cameras[0].width, cameras[0].height), max_views=MAX_VIEW_NUM,
            min_vis=self._min_required_vis_landmarks, crop_size=int(self._input_size[0]),
            focal_multiplier=self._hand_ratio_in_crop, check_indices=False)
        packed = torch.cat([g["crop_params"].reshape(n, -1), g["cam_index"].double(), g["n_views"].double()[:, None],
                            g["status"].double()[:, None]], 1).cpu().numpy()        # one read-back
        crop_cameras: Dict[int, Dict[int, PinholePlaneCameraModel]] = {}
        size = int(self._input_size[0])
        for i, (hand_idx, _pose) in enumerate(hands):
            rows = packed[i, : MAX_VIEW_NUM * 24].reshape(MAX_VIEW_NUM, 24)
            cam_index = packed[i, MAX_VIEW_NUM * 24: MAX_VIEW_NUM * 25].astype(int)
            n_views, status = int(packed[i, -2]), int(packed[i, -1])
            if status != 0:
                raise ValueError("Unable to create crop camera")
            per_hand = {}
            for v in range(n_views):
                t = np.eye(4)
                t[:3, :3] = rows[v, 4:13].reshape(3, 3)
                t[:3, 3] = rows[v, 13:16]
                per_hand[int(cam_index[v])] = PinholePlaneCameraModel(
                    width=size, height=size, f=(rows[v, 0], rows[v, 1]), c=(rows[v, 2], rows[v, 3]), distort_coeffs=[],
                    camera_to_world_xf=t)
            if per_hand and len(per_hand) >= min_num_crops:
                crop_cameras[hand_idx] = per_hand
        return crop_cameras

    # ------------------------------------------------------------------ network inputs
    def _make_inputs(self, sample: InputFrame, hand_model_mm: Optional[HandModel], crop_cameras):
        """Resample every (hand, view) crop on the GPU and assemble the network inputs
        (lib/tracker/tracker.py:315-368).  Dict order defines the sample order."""
        dev = torch.device(self._device)
        if dev.type != "cuda":
            raise _native.NativeLibraryError("HandTracker needs a HIP device: the crop resampler and the network "
                                             "have no CPU fallback")
        used_cams = sorted({ci for per_hand in crop_cameras.values() for ci in per_hand})
        slot_of = {ci: i for i, ci in enumerate(used_cams)}
        src = torch.from_numpy(np.stack([np.ascontiguousarray(sample.views[ci].image) for ci in used_cams])).to(dev)
        cam_rows = np.stack([geometry.pack_camera_model(sample.views[ci].camera) for ci in used_cams])
        crop_rows, src_index, intrinsics, extrinsics, sample_range, hand_indices = [], [], [], [], [], []
        for hand_idx, per_hand in crop_cameras.items():
            start = len(crop_rows)
            for cam_idx, crop_camera in per_hand.items():
                crop_rows.append(geometry.pack_camera_model(crop_camera))
                src_index.append(slot_of[cam_idx])
                k, ext = network_camera_inputs(crop_camera)
                intrinsics.append(k)
                extrinsics.append(ext)
            if len(crop_rows) > start:
                hand_indices.append(hand_idx)
                sample_range.append((start, len(crop_rows)))
        hand_indices = np.array(hand_indices)
        crops = self._model.engine.warp_crops(
            src, torch.from_numpy(cam_rows).to(dev), torch.from_numpy(np.stack(crop_rows)).to(dev),
            torch.tensor(src_index, dtype=torch.int32, device=dev), self._remap_mode)
        frame_data = InputFrameData(
            left_images=crops,
            intrinsics=torch.from_numpy(np.stack(intrinsics)).float().to(dev),
            extrinsics_xf=torch.from_numpy(np.stack(extrinsics)).float().to(dev))
        frame_desc = InputFrameDesc(
            sample_range=torch.tensor(sample_range, dtype=torch.long, device=dev),
            memory_idx=torch.from_numpy(hand_indices).long().to(dev),
            use_memory=torch.from_numpy(self._valid_tracking_history[hand_indices]).bool().to(dev),
            hand_idx=torch.from_numpy(hand_indices).long().to(dev))
        skeleton_data = None
        if hand_model_mm is not None:
            hand_model_m = scaled_hand_model(hand_model_mm, MM_TO_M)
            skeleton_data = InputSkeletonData(
                joint_rotation_axes=hand_model_m.joint_rotation_axes.float().to(dev),
                joint_rest_positions=hand_model_m.joint_rest_positions.float().to(dev))
        return frame_data, frame_desc, skeleton_data

    def _run(self, sample, hand_model, crop_cameras, calibrate: bool) -> TrackingResult:
        if not crop_cameras:
            self.reset_history()       # frame without hands
            return TrackingResult()
        frame_data, frame_desc, skeleton_data = self._make_inputs(sample, hand_model, crop_cameras)
        if calibrate:
            out = self._model.regress_pose_pred_skel_scale(frame_data, frame_desc)
        else:
            out = self._model.regress_pose_use_skeleton(frame_data, frame_desc, skeleton_data)
        return self._gen_tracking_result(out, frame_desc.hand_idx.cpu().numpy(), crop_cameras)

    def track_frame(self, sample: InputFrame, hand_model: HandModel, crop_cameras) -> TrackingResult:
        return self._run(sample, hand_model, crop_cameras, calibrate=False)

    def track_frame_and_calibrate_scale(self, sample: InputFrame, crop_cameras) -> TrackingResult:
        return self._run(sample, None, crop_cameras, calibrate=True)

    def _gen_tracking_result(self, regressor_output: RegressorOutput, hand_indices: np.ndarray, crop_cameras
                             ) -> TrackingResult:
        """m -> mm, per-hand dicts, validity history (lib/tracker/tracker.py:370-412)."""
        ja = regressor_output.joint_angles.to("cpu").numpy()
        xf = regressor_output.wrist_xfs.to("cpu").numpy()
        xf[..., :3, 3] *= M_TO_MM
        scales = None if regressor_output.skel_scales is None else regressor_output.skel_scales.to("cpu").numpy()
        hand_poses, num_views, predicted_scales = {}, {}, {}
        for i, hand_idx in enumerate(hand_indices):
            hand_poses[hand_idx] = SingleHandPose(joint_angles=ja[i], wrist_xform=xf[i], hand_confidence=1.0)
            num_views[hand_idx] = len(crop_cameras[hand_idx])
            if scales is not None:
                predicted_scales[hand_idx] = scales[i]
        for hand_idx in range(NUM_HANDS):
            self._valid_tracking_history[hand_idx] = hand_idx in hand_poses
        return TrackingResult(hand_poses=hand_poses, num_views=num_views, predicted_scales=predicted_scales)
