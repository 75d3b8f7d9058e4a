"""On-disk formats either side of the hot path - host I/O, no arithmetic.

* label JSON of a recording                    lib/tracker/video_pose_data.py:23-96, lib/common/camera.py:423-444
* mono frame -> per-camera views               lib/tracker/video_pose_data.py:128-131,143-156
* `.torch.idx` / `.torch.bin` array files      lib/data_utils/idxbinfile.py:47-189,233-301,368-384
  (msgpack objects for dtype code 8; lib/batched_dataset/sample.py:42-53 consumes them)

Video decoding (PyAV) is out of scope: `SyncedImagePoseStream` takes any iterable of mono frames; asking it to open
an mp4 raises ImportError when PyAV is not installed, exactly like importing the reference's module would.
"""
import enum
import io
import json
import math
from dataclasses import dataclass
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .geometry import CameraModel, read_camera_from_json
from .hand import HandModel


# ----------------------------------------------------------------------------- label JSON
@dataclass
class HandPoseLabels:
    cameras: List[CameraModel]
    camera_angles: List[float]
    camera_to_world_transforms: np.ndarray
    hand_model: HandModel
    joint_angles: np.ndarray
    wrist_transforms: np.ndarray
    hand_confidences: np.ndarray

    def __len__(self):
        return len(self.joint_angles)


def _load_json(p: str):
    with io.open(p, "rb") as bf:
        return json.load(bf)


def load_hand_model_from_dict(hand_model_dict) -> HandModel:
    """Lists become float32 tensors (torch.Tensor(v)), everything else passes through
    (lib/tracker/video_pose_data.py:63-72)."""
    return HandModel(**{k: (torch.Tensor(v) if isinstance(v, list) else v) for k, v in hand_model_dict.items()})


def labels_from_dict(labels: Dict[str, Any]) -> HandPoseLabels:
    return HandPoseLabels(
        cameras=[read_camera_from_json(c) for c in labels["cameras"]],
        camera_angles=labels["camera_angles"],
        camera_to_world_transforms=np.array(labels["camera_to_world_transforms"]),
        hand_model=load_hand_model_from_dict(labels["hand_model"]),
        joint_angles=np.array(labels["joint_angles"]),
        wrist_transforms=np.array(labels["wrist_transforms"]),
        hand_confidences=np.array(labels["hand_confidences"]))


def _load_hand_pose_labels(p: str) -> HandPoseLabels:
    return labels_from_dict(_load_json(p))


def labels_to_arrays(lab: HandPoseLabels) -> Dict[str, np.ndarray]:
    """The flat-array form `pipeline.load_labels` / `crop_plan_*` consume (one row of 14 numbers per camera)."""
    cams = np.array([[c.width, c.height, c.f[0], c.f[1], c.c[0], c.c[1], *tuple(c.distort)] for c in lab.cameras], np.float64)
    out = {"cameras": cams, "camera_angles": np.asarray(lab.camera_angles, np.float64),
           "joint_angles": lab.joint_angles, "wrist_transforms": lab.wrist_transforms,
           "hand_confidences": lab.hand_confidences, "camera_to_world_transforms": lab.camera_to_world_transforms}
    for k in ("joint_rotation_axes", "joint_rest_positions", "landmark_rest_positions", "landmark_rest_bone_weights",
              "landmark_rest_bone_indices", "joint_limits"):
        v = getattr(lab.hand_model, k)
        if v is not None:
            out["hm." + k] = v.numpy()
    return out


# ----------------------------------------------------------------------------- frames
def split_views(raw_mono: np.ndarray, n_cams: int) -> np.ndarray:
    """[H, n_cams*W] mono frame -> [H, n_cams, W] view (no copy), camera c = [:, c, :]
    (lib/tracker/video_pose_data.py:128-131)."""
    return raw_mono.reshape(raw_mono.shape[0], n_cams, -1)


class VideoStream:
    """mp4 -> mono frames through PyAV (lib/tracker/video_pose_data.py:37-55).  PyAV is not part of this image."""

    def __init__(self, data_path: str):
        self._data_path = data_path

    def _open(self):
        try:
            import av
        except ImportError as e:
            raise ImportError("decoding an mp4 needs PyAV (`av`), which is not installed; pass decoded frames to "
                              "SyncedImagePoseStream(frames=...) instead") from e
        return av.open(self._data_path)

    def __len__(self) -> int:
        return self._open().streams.video[0].frames

    def __iter__(self) -> Iterator[np.ndarray]:
        container = self._open()
        stream = container.streams.video[0]
        for frame in container.decode(stream):
            yield np.array(frame.to_image())[..., 0]


class SyncedImagePoseStream:
    """Yields (InputFrame, gt_tracking) per frame like the reference's (lib/tracker/video_pose_data.py:99-153).
    `frames`: optional sequence of decoded mono frames [H, n_cams*W] u8 replacing the mp4 reader."""

    def __init__(self, data_path: str, frames: Optional[Sequence[np.ndarray]] = None):
        self._hand_pose_labels = _load_hand_pose_labels(data_path[:-4] + ".json")
        self._image_stream = VideoStream(data_path) if frames is None else frames
        assert len(self._hand_pose_labels) == len(self._image_stream)

    def __len__(self) -> int:
        return len(self._image_stream)

    def __iter__(self):
        from .tracker import InputFrame, SingleHandPose, ViewData
        lab = self._hand_pose_labels
        for frame_idx, raw_mono in enumerate(self._image_stream):
            gt_tracking = {}
            for hand_idx in range(0, 2):
                if lab.hand_confidences[frame_idx, hand_idx] > 0:
                    gt_tracking[hand_idx] = SingleHandPose(joint_angles=lab.joint_angles[frame_idx, hand_idx],
                                                           wrist_xform=lab.wrist_transforms[frame_idx, hand_idx],
                                                           hand_confidence=lab.hand_confidences[frame_idx, hand_idx])
            views_img = split_views(raw_mono, len(lab.cameras))
            if lab.camera_to_world_transforms[frame_idx].sum() == 0:
                assert not gt_tracking, "Cameras are not tracked, expecting no ground truth tracking!"
            views = [ViewData(image=views_img[:, ci, :],
                              camera=lab.cameras[ci].copy(camera_to_world_xf=lab.camera_to_world_transforms[frame_idx, ci]),
                              camera_angle=lab.camera_angles[ci]) for ci in range(len(lab.cameras))]
            yield InputFrame(views=views), gt_tracking


# ----------------------------------------------------------------------------- .torch.idx / .torch.bin
IDX_MAGIC = 0x584449544E54
OBJECT_DTYPE = np.dtype("object")
MsgpackObject = Dict[str, Any]
RawField = Union[np.ndarray, MsgpackObject]


class BinFormat(enum.Enum):
    TENSOR = 0
    MSGPACK = 1

_CODE_TO_DTYPE = {1: "uint8", 2: "int8", 3: "int16", 4: "int32", 5: "int64", 6: "float32", 7: "float64", 8: "object"}
_DTYPE_TO_CODE = {v: k for k, v in _CODE_TO_DTYPE.items()}
Buffer = Union[bytes, bytearray, memoryview]


def _bin_path_for_idx(idx_path: str) -> str:
    assert idx_path.endswith(".torch.idx")
    return idx_path[: -len(".torch.idx")] + ".torch.bin"


class TorchIdx:
    """Layout of one `.torch.bin` file: an int64 array
    [magic, version, dtype code, itemsize, N, S, N+1 dim offsets, N+1 data offsets (in items), S sizes]
    (lib/data_utils/idxbinfile.py:118-189).  Frames may differ in shape; dtype code 8 = msgpack objects."""

    def __init__(self, path: str, bin_path: Optional[str] = None, buffer: Optional[Buffer] = None) -> None:
        self.source = path
        self.bin_path = _bin_path_for_idx(path) if bin_path is None else bin_path
        if buffer is None:
            with io.open(path, "rb") as f:
                buffer = f.read()
        data = self._as_int64(buffer)
        if data.size < 6:
            raise ValueError(f".torch.idx file {path} is too short")
        if data[1] == 0:
            if data[0] != 0:
                raise ValueError(f"bad magic number in .torch.idx file {path}")
        elif data[1] == 1:
            if data[0] != IDX_MAGIC:
                raise ValueError(f"bad magic number in .torch.idx file {path}")
        else:
            raise ValueError(f"unsupported version {data[1]} in .torch.idx file {path}")
        code = int(data[2])
        if code not in _CODE_TO_DTYPE:
            raise KeyError(f"unrecognized type {code}")
        self.itemsize = int(data[3])
        self.dtype = np.dtype(_CODE_TO_DTYPE[code])
        self._msgpack = code == 8
        if not self._msgpack and self.dtype.itemsize != self.itemsize:
            raise ValueError(f"item size {self.itemsize} not compatible with dtype {self.dtype}.itemsize={self.dtype.itemsize}")
        n, s = int(data[4]), int(data[5])
        if data.size < 6 + 2 * (n + 1) + s:
            raise ValueError(f".torch.idx file {path} is truncated")
        dim_off = data[6: 6 + n + 1]
        data_off = data[6 + n + 1: 6 + 2 * (n + 1)]
        sizes = data[6 + 2 * (n + 1): 6 + 2 * (n + 1) + s]
        self.dims: List[Tuple[int, ...]] = [tuple(int(v) for v in sizes[dim_off[i]: dim_off[i + 1]]) for i in range(n)]
        self._byte_offsets = data_off * self.itemsize
        self.is_uniform = (not self._msgpack) and n > 0 and all(d == self.dims[0] for d in self.dims)
        self.shape: Optional[Tuple[int, ...]] = (n, *self.dims[0]) if self.is_uniform else None

    @staticmethod
    def _as_int64(buffer: Buffer) -> np.ndarray:
        mv = memoryview(buffer)
        if mv.ndim != 1:
            raise ValueError(f".torch.idx data has invalid shape: {mv.shape}; require ndim=1")
        if mv.format == "B":
            if len(mv) % 8 != 0:
                raise ValueError(f".torch.idx data has invalid length: {len(mv)}%8 != 0")
        elif mv.format not in ("q", "l") or mv.itemsize != 8:
            raise ValueError(f".torch.idx data has invalid format {mv.format}: expected 'B' (bytes)  or 'q' or 'l' (int64)")
        return np.frombuffer(mv, dtype=np.int64)

    def __len__(self):
        return len(self.dims)

    def byte_offset(self, i: int) -> int:
        return int(self._byte_offsets[len(self) if i == -1 else i])

    def byte_offsets(self, start: int, end: int) -> np.ndarray:
        """Byte offsets of frames start..end-1, the span form the reference's async reader asks for
        (lib/data_utils/idxbinfile.py:213-231): `end` may be N+1 (the end-of-data offset included); `end == -1` stops
        BEFORE the end-of-data offset like the reference does on both of its branches (`array[start:-1]`, and
        `arange(start, N)`): N - start entries."""
        return self._byte_offsets[start: len(self) if end == -1 else end]

    def data_size_bytes(self) -> int:
        return self.byte_offset(-1) - self.byte_offset(0)

    def item_shape(self, i: Optional[int] = None) -> Tuple[int, ...]:
        if i is None:
            if self.shape is None:
                raise ValueError(f"Dataset does not have uniform shape: {self.source}")
            return self.shape[1:]
        return self.dims[i]

    def _check(self, istart: int, istop: int, buffer: Buffer):
        want = self.byte_offset(istop) - self.byte_offset(istart)
        got = memoryview(buffer).nbytes
        if want != got:
            raise ValueError(f"expected {want} bytes but got {got} for {self.bin_path}[{istart}:{istop}]")

    def view_frame(self, index: int, buffer: Buffer):
        """One frame's bytes -> ndarray view, or the unpacked msgpack object."""
        self._check(index, index + 1, buffer)
        if self._msgpack:
            import msgpack
            return msgpack.unpackb(buffer)
        return np.ndarray(shape=self.dims[index], dtype=self.dtype, buffer=buffer)

    def view_buffer_at(self, index: int, buffer: Buffer):
        base = self.byte_offset(0)
        return self.view_frame(index, memoryview(buffer)[self.byte_offset(index) - base: self.byte_offset(index + 1) - base])

    def view_buffer(self, buffer: Buffer):
        """Whole `.bin` contents -> one [N, ...] array if uniform, else a list of per-frame values."""
        self._check(0, -1, buffer)
        if self.shape is not None:
            return np.ndarray(shape=self.shape, dtype=self.dtype, buffer=buffer)
        return [self.view_buffer_at(i, buffer) for i in range(len(self))]

    def read_bin(self):
        with io.open(self.bin_path, "rb") as f:
            f.seek(self.byte_offset(0))
            return self.view_buffer(f.read(self.data_size_bytes()))


def write_torch_idx_bin(idx_path: str, frames: Union[np.ndarray, Sequence[Any]], bin_path: Optional[str] = None) -> None:
    """Write frames (ndarrays of one dtype, or msgpack-able objects) in the layout TorchIdx parses (version 1)."""
    bin_path = _bin_path_for_idx(idx_path) if bin_path is None else bin_path
    frames = list(frames)
    objects = len(frames) > 0 and not isinstance(frames[0], np.ndarray)
    if objects:
        import msgpack
        blobs = [msgpack.packb(f) for f in frames]
        dims = [(len(b),) for b in blobs]
        code, itemsize = 8, 1
    else:
        arrs = [np.ascontiguousarray(f) for f in frames]
        dt = arrs[0].dtype if arrs else np.dtype("uint8")
        if any(a.dtype != dt for a in arrs):
            raise ValueError("all frames must share one dtype")
        if dt.name not in _DTYPE_TO_CODE or dt.name == "object":
            raise ValueError(f"dtype {dt} has no .torch.idx type code")
        blobs = [a.tobytes() for a in arrs]
        dims = [a.shape for a in arrs]
        code, itemsize = _DTYPE_TO_CODE[dt.name], dt.itemsize
    dim_off = np.concatenate([[0], np.cumsum([len(d) for d in dims])]).astype(np.int64)
    data_off = np.concatenate([[0], np.cumsum([len(b) // itemsize for b in blobs])]).astype(np.int64)
    sizes = np.array([v for d in dims for v in d], np.int64)
    head = np.array([IDX_MAGIC, 1, code, itemsize, len(frames), sizes.size], np.int64)
    with io.open(idx_path, "wb") as f:
        f.write(np.concatenate([head, dim_off, data_off, sizes]).tobytes())
    with io.open(bin_path, "wb") as f:
        for b in blobs:
            f.write(b)


def read_sequence(mono_idx: str, labels_idx: str, index: int) -> Dict[str, Any]:
    """One torch_data sequence as `preprocess` takes it: {"mono": [seq, views, H, W] array, "labels": msgpack dict}
    (run_inference_torch_data.py:133-156 via lib/batched_dataset/sample.py:42-53)."""
    mono, lab = TorchIdx(mono_idx), TorchIdx(labels_idx)
    with io.open(mono.bin_path, "rb") as f:
        f.seek(mono.byte_offset(index))
        m = mono.view_frame(index, f.read(mono.byte_offset(index + 1) - mono.byte_offset(index)))
    with io.open(lab.bin_path, "rb") as f:
        f.seek(lab.byte_offset(index))
        obj = lab.view_frame(index, f.read(lab.byte_offset(index + 1) - lab.byte_offset(index)))
    return {"mono": m, "labels": obj}
