"""Hand model container and forward kinematics entry points with the reference's names
(lib/common/hand.py:11-147, lib/common/hand_skinning.py:189-209).

`skin_landmarks` is part of the hot path: it always runs the HIP
kernel csrc/fk.hip through the C ABI - inputs on the CPU are moved to the GPU and the result is
returned on the input's device.  There is no CPU implementation in the product.
"""
from enum import Enum
from typing import Any, Dict, NamedTuple, Optional

import numpy as np
import torch

from . import _native

NUM_HANDS = 2
NUM_LANDMARKS_PER_HAND = 21
NUM_FINGERTIPS_PER_HAND = 5
NUM_JOINTS_PER_HAND = 22
LEFT_HAND_INDEX = 0
RIGHT_HAND_INDEX = 1
NUM_DIGITS = 5
NUM_JOINT_FRAMES = 1 + 1 + 3 * 5
DOF_PER_FINGER = 4


class LANDMARK(Enum):
    THUMB_FINGERTIP = "Thumb fingertip"
    INDEX_FINGER_FINGERTIP = "Index finger fingertip"
    MIDDLE_FINGER_FINGERTIP = "Middle finger fingertip"
    RING_FINGER_FINGERTIP = "Ring finger fingertip"
    PINKY_FINGER_FINGERTIP = "Pinky finger fingertip"
    WRIST_JOINT = "Wrist joint"
    THUMB_INTERMEDIATE_FRAME = "Thumb intermediate frame"
    THUMB_DISTAL_FRAME = "Thumb distal frame"
    INDEX_PROXIMAL_FRAME = "Index proximal frame"
    INDEX_INTERMEDIATE_FRAME = "Index intermediate frame"
    INDEX_DISTAL_FRAME = "Index distal frame"
    MIDDLE_PROXIMAL_FRAME = "Middle proximal frame"
    MIDDLE_INTERMEDIATE_FRAME = "Middle intermediate frame"
    MIDDLE_DISTAL_FRAME = "Middle distal frame"
    RING_PROXIMAL_FRAME = "Ring proximal frame"
    RING_INTERMEDIATE_FRAME = "Ring intermediate frame"
    RING_DISTAL_FRAME = "Ring distal frame"
    PINKY_PROXIMAL_FRAME = "Pinky proximal frame"
    PINKY_INTERMEDIATE_FRAME = "Pinky intermediate frame"
    PINKY_DISTAL_FRAME = "Pinky distal frame"
    PALM_CENTER = "Palm center"


class HandModel(NamedTuple):
    joint_rotation_axes: torch.Tensor
    joint_rest_positions: torch.Tensor
    joint_frame_index: torch.Tensor
    joint_parent: torch.Tensor
    joint_first_child: torch.Tensor
    joint_next_sibling: torch.Tensor
    landmark_rest_positions: torch.Tensor
    landmark_rest_bone_weights: torch.Tensor
    landmark_rest_bone_indices: torch.Tensor
    hand_scale: Optional[torch.Tensor]
    mesh_vertices: Optional[torch.Tensor] = None
    mesh_triangles: Optional[torch.Tensor] = None
    dense_bone_weights: Optional[torch.Tensor] = None
    joint_limits: Optional[torch.Tensor] = None

    @classmethod
    def from_json(cls, json_data: Dict[str, Any]) -> "HandModel":
        return cls(**{k: (torch.tensor(v) if v is not None else None) for k, v in json_data.items()})

    def to_json(self) -> Dict[str, Any]:
        return {k: (v.tolist() if isinstance(v, torch.Tensor) else v) for k, v in self._asdict().items()}


def _lead_factor(hand: HandModel, multiplier) -> torch.Tensor:
    lead = hand.joint_rest_positions.shape[:-2]
    ones = torch.ones(lead, dtype=hand.joint_rest_positions.dtype, device=hand.joint_rest_positions.device)
    return (ones * multiplier)[..., None, None]


def scaled_hand_model(hand: HandModel, multiplier) -> HandModel:
    """Scale every length of the model (lib/common/hand.py:78-111)."""
    m = _lead_factor(hand, multiplier)
    return hand._replace(
        joint_rest_positions=hand.joint_rest_positions * m,
        landmark_rest_positions=hand.landmark_rest_positions * m,
        mesh_vertices=None if hand.mesh_vertices is None else hand.mesh_vertices * m,
    )


def mirrored_hand_model(hand: HandModel, to_mirror: torch.Tensor) -> HandModel:
    """Left<->right mirror of the selected models: x of positions and y,z of rotation axes change sign
    (lib/common/hand.py:114-147; mesh vertices are left untouched there too)."""
    sel = to_mirror.reshape(-1).to(torch.bool)
    lead = to_mirror.shape

    def flip(t: torch.Tensor, cols: slice) -> torch.Tensor:
        out = t.clone()
        flat = out.reshape((-1,) + t.shape[len(lead):])
        sub = flat[sel]
        sub[..., cols] = -sub[..., cols]
        flat[sel] = sub
        return flat.reshape(t.shape)

    return hand._replace(
        joint_rotation_axes=flip(hand.joint_rotation_axes, slice(1, None)),
        joint_rest_positions=flip(hand.joint_rest_positions, slice(0, 1)),
        landmark_rest_positions=flip(hand.landmark_rest_positions, slice(0, 1)),
        mesh_vertices=None if hand.mesh_vertices is None else hand.mesh_vertices.clone(),
    )


_fk_engine_handles: Dict[int, "_native.HipEngine"] = {}


def fk_device() -> torch.device:
    if not torch.cuda.is_available():
        raise _native.NativeLibraryError(
            "skin_landmarks runs on the HIP kernel csrc/fk.hip and no HIP device is visible "
            "(there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


_BLOB_FIELDS = ("joint_rotation_axes", "joint_rest_positions", "landmark_rest_positions", "landmark_rest_bone_weights",
                "landmark_rest_bone_indices")
_blob_cache: list = []      # [(key, tensors kept alive, device blob)], most recent first


def device_blob(hand_model: HandModel, dev: torch.device) -> torch.Tensor:
    """The packed [n_models, 321] fp32 model the FK / crop-camera kernels read, cached on the device: the per-frame
    API skins the same HandModel several times per frame.  Keyed on tensor identity + in-place version counters."""
    tensors = tuple(getattr(hand_model, f) for f in _BLOB_FIELDS)
    key = (str(dev),) + tuple((id(t), t._version) for t in tensors)
    for i, (k, _keep, blob) in enumerate(_blob_cache):
        if k == key:
            if i:
                _blob_cache.insert(0, _blob_cache.pop(i))
            return blob
    blob = torch.from_numpy(_native.hand_model_blob(*tensors).reshape(-1, 321)).to(dev)
    _blob_cache.insert(0, (key, tensors, blob))
    del _blob_cache[8:]
    return blob


def skin_landmarks(hand_model: HandModel, joint_angles: torch.Tensor, wrist_transforms: torch.Tensor) -> torch.Tensor:
    """[...,22] joint angles + [...,4,4] wrist transforms -> [...,21,3] landmarks, any leading dims; the
    model's tensors are either unbatched or carry the same leading dims (lib/common/hand_skinning.py:189-209)."""
    lead = tuple(joint_angles.shape[:-1])
    n = int(np.prod(lead)) if lead else 1
    model_lead = tuple(hand_model.joint_rest_positions.shape[:-2])
    if model_lead not in ((), lead):
        raise AssertionError(f"Leading dimensions do not match, got {lead} and {model_lead}")
    src_device = joint_angles.device
    dev = src_device if src_device.type == "cuda" else fk_device()
    out = _native.fk_stateless(device_blob(hand_model, dev),
                               joint_angles.reshape(n, 22).to(dev, torch.float32),
                               wrist_transforms.reshape(n, 4, 4).to(dev, torch.float32))
    return out.reshape(lead + (21, 3)).to(src_device)
