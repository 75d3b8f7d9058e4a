"""UmeTrackModel with the reference's call surface, executing on libumetrack_hip.so.

Mirrors lib/models/umetrack_model.py:21-242 (input/output dataclasses and the two regress
entry points), lib/models/regressor.py:124-129 (RegressorOutput) and
lib/models/model_loader.py:53-88 (load_pretrained_model).  The object holds the weights as a
reference-keyed state dict; `.to("cuda")` packs them into a native handle (BatchNorm folded,
MFMA layouts) - all arithmetic of the forward pass happens in the HIP kernels.  On a machine
without a HIP device the regress methods raise: there is no CPU path.
"""
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _native, arch


@dataclass
class InputFrameData:
    """left_images [n_images,96,96], intrinsics [n_images,3,3], extrinsics_xf [n_images,4,4]."""
    left_images: torch.Tensor
    intrinsics: torch.Tensor
    extrinsics_xf: torch.Tensor


@dataclass
class InputFrameDesc:
    """sample_range [bs,2] (first,last+1 image of each sample), memory_idx [bs], use_memory [bs] bool,
    hand_idx [bs]."""
    sample_range: torch.Tensor
    memory_idx: torch.Tensor
    use_memory: torch.Tensor
    hand_idx: torch.Tensor


@dataclass
class InputSkeletonData:
    """joint_rotation_axes / joint_rest_positions: [22,3] (shared) or [bs,22,3], metres."""
    joint_rotation_axes: torch.Tensor
    joint_rest_positions: torch.Tensor


@dataclass
class RegressorOutput:
    joint_angles: torch.Tensor
    wrist_xfs: torch.Tensor
    skel_scales: Optional[torch.Tensor] = None
    landmark_uncertainty_sigmas: Optional[torch.Tensor] = None


class UmeTrackModel:
    """Weights + native engine.  Not an nn.Module: nothing here is differentiable or runs in ATen."""

    def __init__(self, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        self._state: Dict[str, torch.Tensor] = {}
        self._device = torch.device("cpu")
        self._engine: Optional[_native.HipEngine] = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ------------------------------------------------------------------ nn.Module-like surface
    def load_state_dict(self, state_dict, strict: bool = True):
        spec = arch.state_dict_spec()
        keys = [k for k, _s, _kind in spec]
        missing = [k for k in keys if k not in state_dict]
        unexpected = [k for k in state_dict if k not in set(keys)]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for UmeTrackModel: "
                               f"Missing key(s): {missing[:6]} Unexpected key(s): {unexpected[:6]}")
        new = {}
        for k, shape, _kind in spec:
            if k not in state_dict:
                if k in self._state:
                    new[k] = self._state[k]
                continue
            t = state_dict[k]
            t = torch.from_numpy(np.asarray(t)) if not isinstance(t, torch.Tensor) else t.detach().cpu()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {tuple(shape)}")
            new[k] = t.clone()
        self._state = new
        self._drop_engine()
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._state)

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("UmeTrackModel is inference-only (BatchNorm is folded into the packed weights)")
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else device
        if device != self._device:
            self._drop_engine()
            self._device = device
        return self

    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    @property
    def engine(self) -> _native.HipEngine:
        if self._engine is None:
            if self._device.type != "cuda":
                raise _native.NativeLibraryError(
                    f"UmeTrackModel is on {self._device}: the forward pass runs only on a HIP device "
                    "(model.to('cuda')); there is no CPU fallback")
            if len(self._state) != len(arch.state_dict_spec()):
                raise RuntimeError("UmeTrackModel has no weights: call load_state_dict first")
            self._engine = _native.HipEngine(self._state, self._device)
        return self._engine

    def getInputImageSizes(self) -> Tuple[int, int]:
        return (arch.CROP, arch.CROP)

    def reset_temporal_memory(self):
        """A fresh temporal state, as a newly constructed SimpleConvRNN (lib/models/temporal.py:40-41)."""
        if self._engine is not None:
            self._engine.reset_memory()

    # ------------------------------------------------------------------ forward
    def _regress(self, frame_data: InputFrameData, frame_desc: InputFrameDesc,
                 skel_data: Optional[InputSkeletonData], mode: int) -> RegressorOutput:
        eng = self.engine
        dev = eng.device
        images = frame_data.left_images
        n, s = images.shape[0], frame_desc.sample_range.shape[0]
        all_multiview = (2 * s == n)
        if mode == _native.UT_MODE_UNKNOWN and not all_multiview:
            raise AssertionError("Unsupported: found single-view samples when calibration scale")
        skel = None
        if skel_data is not None:
            axes = skel_data.joint_rotation_axes.to(dev, torch.float32).reshape(-1, arch.N_JOINTS, 3)
            rest = skel_data.joint_rest_positions.to(dev, torch.float32).reshape(-1, arch.N_JOINTS, 3)
            skel = torch.stack([axes, rest], dim=1).contiguous()
            if skel.shape[0] not in (1, s):
                raise ValueError(f"skeleton batch {skel.shape[0]} does not match {s} samples")
        # lib/models/temporal.py:102 - the reference also reads max(memory_idx) back to the host
        n_slots = int(frame_desc.memory_idx.max()) + 1
        feat = eng.backbone(images.to(dev))
        pose, _ = eng.fuse_temporal_regress(
            feat, frame_data.intrinsics.to(dev), frame_data.extrinsics_xf.to(dev), frame_desc.sample_range.to(dev),
            frame_desc.memory_idx.to(dev), frame_desc.use_memory.to(dev), frame_desc.hand_idx.to(dev), n_slots,
            all_multiview, skel, mode)
        return RegressorOutput(
            joint_angles=pose[:, 0:22].contiguous(),
            wrist_xfs=pose[:, 22:38].reshape(s, 4, 4),
            skel_scales=pose[:, 38].contiguous() if mode == _native.UT_MODE_UNKNOWN else None,
            landmark_uncertainty_sigmas=pose[:, 39:60].contiguous())

    def regress_pose_use_skeleton(self, frame_data: InputFrameData, frame_desc: InputFrameDesc,
                                  skel_data: InputSkeletonData) -> RegressorOutput:
        return self._regress(frame_data, frame_desc, skel_data, _native.UT_MODE_KNOWN)

    def regress_pose_pred_skel_scale(self, frame_data: InputFrameData, frame_desc: InputFrameDesc) -> RegressorOutput:
        return self._regress(frame_data, frame_desc, None, _native.UT_MODE_UNKNOWN)


def load_pretrained_model(model_path: str) -> UmeTrackModel:
    """Read a plain state_dict checkpoint (lib/models/model_loader.py:84-87).  weights_only=True: nothing
    from the file is executed."""
    with open(model_path, "rb") as fp:
        sd = torch.load(fp, map_location="cpu", weights_only=True)
    return UmeTrackModel(sd)
