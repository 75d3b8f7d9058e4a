"""The torch_data batch path: the other producer of the view-stacked crop tensor.

Host mirror of lib/batched_dataset/{sample,data_transform}.py and of the batching helpers of
run_inference_torch_data.py:39-130, with the per-view crop matrices and the pinhole->pinhole resampler running on
the GPU (ut_gen_crop_matrices / ut_resample_homography): one launch each per sequence batch instead of a Python
loop over frames x views and a numpy scatter.  Same names, argument meaning and error behaviour as the reference.
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _native, bundles
from .hand import HandModel, mirrored_hand_model, scaled_hand_model, skin_landmarks
from .model import InputFrameData, InputFrameDesc, InputSkeletonData

scalar_type = np.float32


# ----------------------------------------------------------------------------- lib/batched_dataset/sample.py
@dataclass
class RawSample:
    images: np.ndarray
    extrinsics: np.ndarray
    intrinsics: np.ndarray
    enclosing_points: np.ndarray
    hand: np.ndarray
    hand_model: HandModel
    wrist: np.ndarray
    joint_angles: np.ndarray
    solved_wrist_xfs: np.ndarray
    solved_joint_angles: np.ndarray
    generic_hand_model: HandModel
    pinch: np.ndarray

    def scaled(self, factor: float):
        """In-place unit change of every length (lib/batched_dataset/sample.py:33-39)."""
        self.extrinsics[..., :3, 3] *= factor
        self.enclosing_points *= factor
        self.wrist[..., :3, 3] *= factor
        self.solved_wrist_xfs[..., :3, 3] *= factor
        self.hand_model = scaled_hand_model(self.hand_model, factor)
        self.generic_hand_model = scaled_hand_model(self.generic_hand_model, factor)


def parse_raw_buffers(mono: np.ndarray, labels: Dict[str, Any]) -> RawSample:
    """lib/batched_dataset/sample.py:42-53: msgpack label dict + image block -> typed sample."""
    typed = {}
    for field, value in labels.items():
        if "hand_model" in field:
            typed[field] = HandModel(**{k: torch.tensor(v) for k, v in value.items()})
        else:
            typed[field] = np.array(value, dtype=np.float32)
    return RawSample(**{"images": mono, **typed})


# ----------------------------------------------------------------------------- lib/batched_dataset/data_transform.py
@dataclass
class PoseData:
    joint_angles: torch.Tensor
    wrist_xfs: torch.Tensor
    left_hand_model: HandModel


@dataclass
class ModelInput:
    orig_pose_data: PoseData
    s_solved_pose_data: PoseData
    left_images: torch.Tensor
    intrinsics: torch.Tensor
    extrinsics_xf: torch.Tensor
    hand_idx: torch.Tensor


@dataclass
class PerBranchOutput:
    joint_angles: torch.Tensor
    wrist_xfs: torch.Tensor
    skel_scales: Optional[torch.Tensor] = None
    pinch_prediction: Optional[torch.Tensor] = None


@dataclass
class ModelTarget:
    gt_skel_targets: PerBranchOutput
    preds_targets: PerBranchOutput
    intrinsics: Optional[torch.Tensor] = None
    extrinsics_xf: Optional[torch.Tensor] = None


def _device(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise _native.NativeLibraryError("the torch_data crop path runs only on a HIP device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _perspective_crop_images(orig_images, orig_extrinsics, orig_intrinsics, crop_points, hand_idx: int,
                             crop_size: Tuple[int, int], device=None, keep_on_device: bool = False):
    """lib/batched_dataset/data_transform.py:215-283.  orig_images [frames, views, H, W] (u8 or f32; numpy or a
    tensor already on the GPU), extrinsics [frames, views, 4, 4] world->eye, intrinsics [frames, views, 3, 3],
    crop_points [frames, pts, 3] -> [crops in [0,1] f32 [frames, views, h, w], extrinsics_xf, new_intrinsics]."""
    if crop_size[0] != crop_size[1]:
        raise ValueError("square crops only")
    dev = _device(device)
    as_dev = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, dt)
    img = orig_images if isinstance(orig_images, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(orig_images))
    img = img.to(dev)
    if img.dtype not in (torch.uint8, torch.float32):
        img = img.float()
    n_frames, n_views = img.shape[:2]
    hand = torch.full((n_frames,), int(hand_idx), dtype=torch.int64, device=dev)
    m = _native.gen_crop_matrices(as_dev(orig_extrinsics, torch.float32), as_dev(orig_intrinsics, torch.float32),
                                  as_dev(crop_points, torch.float32), hand, crop_size=int(crop_size[0]))
    if bool((m["status"] != 0).any()):
        raise ValueError("Unable to create crop camera")
    crops = _native.resample_homography(img.reshape(-1, *img.shape[2:]), m["resample_xf"].reshape(-1, 4, 4),
                                        (int(crop_size[1]), int(crop_size[0])))
    crops = crops.reshape(n_frames, n_views, int(crop_size[1]), int(crop_size[0]))
    out = [crops, m["extrinsics_xf"], m["new_intrinsics"]]
    return out if keep_on_device else [t.cpu().numpy() for t in out]


def prepare_inputs_targets(sample: RawSample, crop_size: Tuple[int, int], device=None) -> Tuple[ModelInput, ModelTarget]:
    """lib/batched_dataset/data_transform.py:286-384."""
    def to_th(t_in: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(t_in).float()

    sample.scaled(0.001)          # mm -> m
    seq_length = sample.images.shape[0]

    def repeat_seq_length(t_in: torch.Tensor) -> torch.Tensor:
        return t_in.unsqueeze(0).expand(seq_length, *t_in.shape)

    generic = bundles.map_fields(repeat_seq_length, sample.generic_hand_model, only_type=torch.Tensor)
    left_generic = mirrored_hand_model(generic, to_th(sample.hand) == 1)
    own = bundles.map_fields(repeat_seq_length, sample.hand_model, only_type=torch.Tensor)
    left_own = mirrored_hand_model(own, to_th(sample.hand) == 1)
    solved = PoseData(wrist_xfs=to_th(sample.solved_wrist_xfs), joint_angles=to_th(sample.solved_joint_angles),
                      left_hand_model=left_generic)
    orig = PoseData(wrist_xfs=to_th(sample.wrist), joint_angles=to_th(sample.joint_angles), left_hand_model=left_own)
    left_images, extrinsics_xf, intrinsics = _perspective_crop_images(
        sample.images, sample.extrinsics, sample.intrinsics, sample.enclosing_points, int(sample.hand[0]), crop_size,
        device=device)
    model_input = ModelInput(orig_pose_data=orig, s_solved_pose_data=solved, left_images=to_th(left_images),
                             intrinsics=to_th(intrinsics), extrinsics_xf=to_th(extrinsics_xf), hand_idx=to_th(sample.hand))
    gt = PerBranchOutput(joint_angles=orig.joint_angles, wrist_xfs=orig.wrist_xfs,
                         skel_scales=orig.left_hand_model.hand_scale, pinch_prediction=to_th(sample.pinch))
    preds = PerBranchOutput(joint_angles=solved.joint_angles, wrist_xfs=solved.wrist_xfs,
                            skel_scales=solved.left_hand_model.hand_scale, pinch_prediction=to_th(sample.pinch))
    target = ModelTarget(gt_skel_targets=gt, preds_targets=preds, intrinsics=to_th(intrinsics),
                         extrinsics_xf=to_th(extrinsics_xf))
    return model_input, target


def preprocess(data: Dict[str, Any], crop_size: Tuple[int, int]) -> Tuple[ModelInput, ModelTarget]:
    """lib/batched_dataset/data_transform.py:387-397; `data` has keys "mono" and "labels"."""
    return prepare_inputs_targets(parse_raw_buffers(**data), crop_size)


# ----------------------------------------------------------------------------- run_inference_torch_data.py:39-130
def unpack_batched_data(training_input: ModelInput, seq_mode: str
                        ) -> List[Tuple[InputFrameData, InputFrameDesc, InputSkeletonData]]:
    """[bs, seq, views, ...] batch -> one (frame_data, frame_desc, skel_data) per time step: slot b of the temporal
    memory belongs to sequence b, memory is used from the second step on (run_inference_torch_data.py:39-85)."""
    if seq_mode == "multiv":
        nv = 2
    elif seq_mode == "singlev":
        nv = 1
    else:
        raise ValueError(f"Unknown sequence mode: {seq_mode}")
    left_images = training_input.left_images
    bs, seq_len = left_images.shape[0], left_images.shape[1]
    dev = left_images.device
    hm = training_input.orig_pose_data.left_hand_model
    sample_range = torch.tensor([(i * nv, (i + 1) * nv) for i in range(bs)], device=dev).long()
    steps = []
    for i_frame in range(seq_len):
        use_memory = torch.ones(bs, device=dev, dtype=torch.bool)
        if i_frame == 0:
            use_memory[:] = False
        frame_data = InputFrameData(
            left_images=torch.flatten(left_images[:, i_frame, 0:nv], 0, 1),
            intrinsics=torch.flatten(training_input.intrinsics[:, i_frame, 0:nv], 0, 1),
            extrinsics_xf=torch.flatten(training_input.extrinsics_xf[:, i_frame, 0:nv], 0, 1))
        frame_desc = InputFrameDesc(hand_idx=training_input.hand_idx[:, i_frame].long(), sample_range=sample_range,
                                    memory_idx=torch.arange(0, bs, device=dev).long(), use_memory=use_memory)
        skel_data = InputSkeletonData(joint_rotation_axes=hm.joint_rotation_axes[:, i_frame],
                                      joint_rest_positions=hm.joint_rest_positions[:, i_frame])
        steps.append((frame_data, frame_desc, skel_data))
    return steps


def eval_batch_keypoints(model, model_input: ModelInput, model_target: ModelTarget, cur_mode: str, use_skel: bool, device
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run a collated batch through the model step by step; (gt_keypoints, output_keypoints) [bs, seq, 21, 3] in
    metres (run_inference_torch_data.py:88-130)."""
    hand_model = mirrored_hand_model(model_input.orig_pose_data.left_hand_model, model_input.hand_idx == 1)
    outputs = []
    for frame_data, frame_desc, skel_input in unpack_batched_data(model_input, cur_mode):
        frame_data, frame_desc, skel_input = bundles.to_device((frame_data, frame_desc, skel_input), device)
        if use_skel:
            cur = model.regress_pose_use_skeleton(frame_data, frame_desc, skel_input)
        else:
            assert cur_mode == "multiv", "Skeleton scale prediction requires multiv data"
            cur = model.regress_pose_pred_skel_scale(frame_data, frame_desc)
        outputs.append(bundles.to_device(cur, torch.device("cpu")))
    batched = bundles.collate(outputs)
    batched = bundles.map_fields(lambda t: t.transpose(0, 1) if t is not None else None, batched)
    target = model_target.gt_skel_targets
    gt_keypoints = skin_landmarks(hand_model, target.joint_angles, target.wrist_xfs)
    output_keypoints = skin_landmarks(hand_model, batched.joint_angles, batched.wrist_xfs)
    return gt_keypoints, output_keypoints


def _eval_batch(model, model_input: ModelInput, model_target: ModelTarget, cur_mode: str, use_skel: bool, device
                ) -> torch.Tensor:
    """Per-sequence mean keypoint error in mm [bs] (run_inference_torch_data.py:88-135)."""
    gt_keypoints, output_keypoints = eval_batch_keypoints(model, model_input, model_target, cur_mode, use_skel, device)
    return (gt_keypoints - output_keypoints).norm(dim=-1).mean(dim=(1, 2)) * 1000
