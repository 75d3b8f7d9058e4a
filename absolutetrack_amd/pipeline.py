"""Batched hot path: many independent frames per launch sequence, sharded across GPUs.

The reference drives the path one frame at a time (run_eval_known_skeleton.py:68-89) and its only
parallelism is one process per recording (`Pool(8)`, :117-119).  Frames are independent, so the
MI355X-native shape of the same computation is: stack the frames of a shard on one GPU, run each
stage ONCE over the whole stack (resample -> backbone -> fuse/temporal/regress -> FK), and shard
contiguous frame blocks across the ranks of a node with a single all-gather of the packed per-hand
records at the end (SURVEY.md section 8 e).  No collective sits inside the data path.

record layout (fp32, 123 per hand-frame): pose record [60] (ut_fuse_temporal_regress) | 21x3 keypoints (mm)
"""
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native, arch, geometry
from .hand import HandModel
from .tracker import (HandTrackerOpts, MAX_VIEW_NUM, SingleHandPose, gen_crop_cameras_from_pose,
                      network_camera_inputs)

RECORD = arch.POSE_REC + arch.N_LANDMARKS * 3
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "recording_00_labels.npz")
_CAM_FIELDS = ("ImageSizeX", "ImageSizeY", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "k5", "k6")


# ----------------------------------------------------------------------------- label data
def load_labels(path: str = _DATA) -> Dict[str, np.ndarray]:
    """sample_data/recording_00.json of the reference as arrays (written by oracle/gen_goldens.py)."""
    return dict(np.load(path))


def hand_model_from_labels(lab: Dict[str, np.ndarray]) -> HandModel:
    t = {k[3:]: torch.from_numpy(v) for k, v in lab.items() if k.startswith("hm.")}
    z = torch.zeros(22)
    return HandModel(joint_rotation_axes=t["joint_rotation_axes"], joint_rest_positions=t["joint_rest_positions"],
                     joint_frame_index=z, joint_parent=z, joint_first_child=z, joint_next_sibling=z,
                     landmark_rest_positions=t["landmark_rest_positions"],
                     landmark_rest_bone_weights=t["landmark_rest_bone_weights"],
                     landmark_rest_bone_indices=t["landmark_rest_bone_indices"], hand_scale=None,
                     joint_limits=t["joint_limits"])


def cameras_for_frame(lab: Dict[str, np.ndarray], frame: int) -> List[geometry.Fisheye62CameraModel]:
    cams = []
    for ci in range(lab["cameras"].shape[0]):
        js = dict(zip(_CAM_FIELDS, lab["cameras"][ci]))
        js["DistortionModel"] = "FishEye62"
        js["ImageSizeX"], js["ImageSizeY"] = int(js["ImageSizeX"]), int(js["ImageSizeY"])
        cams.append(geometry.read_camera_from_json(js).copy(camera_to_world_xf=lab["camera_to_world_transforms"][frame, ci]))
    return cams


# ----------------------------------------------------------------------------- batch description
@dataclass
class FrameBatch:
    """Device-resident inputs of one pass over F frames (S hand-samples, N crops)."""
    src: torch.Tensor            # u8  [F*C, H, W]
    cam_params: torch.Tensor     # f64 [F*C, 32]
    crop_params: torch.Tensor    # f64 [N, 24]
    src_index: torch.Tensor      # i32 [N]
    intrinsics: torch.Tensor     # f32 [N,3,3]
    extrinsics: torch.Tensor     # f32 [N,4,4]
    sample_range: torch.Tensor   # i64 [S,2]
    memory_idx: torch.Tensor     # i64 [S]
    use_memory: torch.Tensor     # u8  [S]
    hand_idx: torch.Tensor       # i64 [S]
    n_slots: int
    all_multiview: bool

    @property
    def n_samples(self) -> int:
        return self.sample_range.shape[0]

    @property
    def n_crops(self) -> int:
        return self.crop_params.shape[0]


def crop_plan_from_labels(lab: Dict[str, np.ndarray], hand_model: HandModel, frame_ids: Sequence[int],
                          hands: Sequence[int] = (0, 1), opts: Optional[HandTrackerOpts] = None):
    """Crop cameras for every (frame, hand) exactly as HandTracker.gen_crop_cameras picks them
    (lib/tracker/tracker.py:222-260), as flat numpy rows.  Label frames repeat with period len(labels), so
    the per-label-frame result is computed once and tiled."""
    opts = opts or HandTrackerOpts()
    n_lab = lab["joint_angles"].shape[0]
    cache: Dict[int, list] = {}
    cam_rows, crop_rows, src_index, intr, extr, ranges, hand_idx = [], [], [], [], [], [], []
    n_cams = lab["cameras"].shape[0]
    for f_out, f in enumerate(frame_ids):
        lf = int(f) % n_lab
        if lf not in cache:
            cams = cameras_for_frame(lab, lf)
            entry = {"cams": [geometry.pack_camera_model(c) for c in cams], "hands": {}}
            for h in hands:
                if lab["hand_confidences"][lf, h] < 0.5:
                    continue
                pose = SingleHandPose(joint_angles=lab["joint_angles"][lf, h], wrist_xform=lab["wrist_transforms"][lf, h],
                                      hand_confidence=float(lab["hand_confidences"][lf, h]))
                cc = gen_crop_cameras_from_pose(cams, lab["camera_angles"], hand_model, pose, h, opts.num_crop_points,
                                                np.array([arch.CROP, arch.CROP]), max_view_num=MAX_VIEW_NUM,
                                                sort_camera_index=True, focal_multiplier=opts.hand_ratio_in_crop,
                                                mirror_right_hand=True,
                                                min_required_vis_landmarks=opts.min_required_vis_landmarks)
                if cc:
                    entry["hands"][h] = [(ci, geometry.pack_camera_model(c), *network_camera_inputs(c)) for ci, c in cc.items()]
            cache[lf] = entry
        entry = cache[lf]
        cam_rows.extend(entry["cams"])
        for h, views in entry["hands"].items():
            start = len(crop_rows)
            for ci, row, k, ext in views:
                crop_rows.append(row)
                src_index.append(f_out * n_cams + ci)
                intr.append(k)
                extr.append(ext)
            ranges.append((start, len(crop_rows)))
            hand_idx.append(h)
    return {"cam_params": np.stack(cam_rows), "crop_params": np.stack(crop_rows),
            "src_index": np.asarray(src_index, np.int32), "intrinsics": np.stack(intr).astype(np.float32),
            "extrinsics": np.stack(extr).astype(np.float32), "sample_range": np.asarray(ranges, np.int64),
            "hand_idx": np.asarray(hand_idx, np.int64)}


def label_candidates(lab: Dict[str, np.ndarray], frame_ids: Sequence[int], hands: Sequence[int] = (0, 1)):
    """Flat (frame, hand) candidates with confidence >= 0.5 and the per-frame camera rows, as numpy arrays:
    the inputs of ut_gen_crop_cameras for these frames (lib/tracker/tracker.py:236-241 confidence gate)."""
    n_lab = lab["joint_angles"].shape[0]
    lf = np.asarray(frame_ids, np.int64) % n_lab
    n_cams = lab["cameras"].shape[0]
    intr = lab["cameras"][:, 2:14]                                       # fx fy cx cy k1..k6 (pack order)
    cam = np.zeros((len(lf), n_cams, 32), np.float64)
    cam[:, :, 0:12] = intr[None]
    c2w = lab["camera_to_world_transforms"][lf]
    cam[:, :, 12:21] = c2w[:, :, :3, :3].reshape(len(lf), n_cams, 9)
    cam[:, :, 21:24] = c2w[:, :, :3, 3]
    conf = lab["hand_confidences"][lf][:, list(hands)] >= 0.5
    fi, hi = np.nonzero(conf)
    hand = np.asarray(hands, np.int64)[hi]
    return {"cam_params": cam.reshape(-1, 32), "frame_idx": fi.astype(np.int32), "hand_idx": hand,
            "joint_angles": lab["joint_angles"][lf[fi], hand].astype(np.float32),
            "wrist_xf": lab["wrist_transforms"][lf[fi], hand].astype(np.float32),
            "camera_angles": np.asarray(lab["camera_angles"], np.float64),
            "src_wh": (int(lab["cameras"][0, 0]), int(lab["cameras"][0, 1])), "n_cams": n_cams}


def crop_plan_on_device(lab: Dict[str, np.ndarray], hand_model: HandModel, frame_ids: Sequence[int], device,
                        hands: Sequence[int] = (0, 1), opts: Optional[HandTrackerOpts] = None) -> Dict[str, torch.Tensor]:
    """crop_plan_from_labels with the per-(frame, hand) geometry done by ONE ut_gen_crop_cameras launch
    (SURVEY.md section 8 row f1).  Returns device tensors with the same keys and row order.  Hands without an
    eligible view are dropped like the host path drops them; a candidate the reference would raise on
    ("Unable to create crop camera") raises here as well."""
    opts = opts or HandTrackerOpts()
    dev = torch.device(device)
    c = label_candidates(lab, frame_ids, hands)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    blob = torch.from_numpy(_native.hand_model_blob(
        hand_model.joint_rotation_axes, hand_model.joint_rest_positions, hand_model.landmark_rest_positions,
        hand_model.landmark_rest_bone_weights, hand_model.landmark_rest_bone_indices)).reshape(1, 321).to(dev)
    cam_params, hand_idx, frame_idx = t(c["cam_params"]), t(c["hand_idx"]), t(c["frame_idx"])
    g = _native.gen_crop_cameras(cam_params, t(c["camera_angles"]), blob, hand_model.joint_limits.float().to(dev),
                                 t(c["joint_angles"]), t(c["wrist_xf"]), frame_idx, hand_idx, c["n_cams"], c["src_wh"],
                                 max_views=MAX_VIEW_NUM, min_vis=opts.min_required_vis_landmarks, crop_size=arch.CROP,
                                 focal_multiplier=opts.hand_ratio_in_crop)
    if bool((g["status"] != 0).any()):
        raise ValueError("Unable to create crop camera")
    nv = g["n_views"].long()
    keep = nv > 0
    used = g["cam_index"] >= 0                                           # [n,V], view slots are filled front to back
    ends = torch.cumsum(nv, 0)
    ranges = torch.stack([ends - nv, ends], 1)[keep]
    src_index = (frame_idx.long()[:, None] * c["n_cams"] + g["cam_index"].long())[used].int()
    return {"cam_params": cam_params, "crop_params": g["crop_params"][used], "src_index": src_index,
            "intrinsics": g["intrinsics"][used], "extrinsics": g["extrinsics"][used], "sample_range": ranges,
            "hand_idx": hand_idx[keep]}


class DeviceCropPlanner:
    """Row f1 inside the step: the label poses of a frame block stay on the GPU and every call regenerates the
    crop cameras with one ut_gen_crop_cameras launch and no host round trip, like the reference's per-frame loop
    calls gen_crop_cameras before track_frame (run_eval_known_skeleton.py:70-81).  The padded [S, 2] output is the
    compact one when every candidate has two views; `ok` (a device flag) says whether that held and no candidate
    was unbuildable - read it after the timed region."""

    def __init__(self, lab: Dict[str, np.ndarray], hand_model: HandModel, frame_ids: Sequence[int], device,
                 opts: Optional[HandTrackerOpts] = None):
        self.opts = opts or HandTrackerOpts()
        dev = torch.device(device)
        c = label_candidates(lab, frame_ids)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.n_cams, self.src_wh = c["n_cams"], c["src_wh"]
        self.cam_params, self.camera_angles = t(c["cam_params"]), t(c["camera_angles"])
        self.joint_angles, self.wrist_xf = t(c["joint_angles"]), t(c["wrist_xf"])
        self.frame_idx, self.hand_idx = t(c["frame_idx"]), t(c["hand_idx"])
        self.blob = torch.from_numpy(_native.hand_model_blob(
            hand_model.joint_rotation_axes, hand_model.joint_rest_positions, hand_model.landmark_rest_positions,
            hand_model.landmark_rest_bone_weights, hand_model.landmark_rest_bone_indices)).reshape(1, 321).to(dev)
        self.limits = hand_model.joint_limits.float().to(dev)
        if int(self.frame_idx.max()) * self.n_cams + self.n_cams > self.cam_params.shape[0]:
            raise ValueError("frame_idx points past cam_params")
        self._src_base = self.frame_idx.long()[:, None] * self.n_cams
        self.ok = torch.ones((), dtype=torch.bool, device=dev)

    def refresh(self, batch: FrameBatch) -> FrameBatch:
        """Overwrite the crop-camera tensors of an all-two-view batch in place from a fresh launch."""
        g = _native.gen_crop_cameras(self.cam_params, self.camera_angles, self.blob, self.limits, self.joint_angles,
                                     self.wrist_xf, self.frame_idx, self.hand_idx, self.n_cams, self.src_wh,
                                     max_views=MAX_VIEW_NUM, min_vis=self.opts.min_required_vis_landmarks,
                                     crop_size=arch.CROP, focal_multiplier=self.opts.hand_ratio_in_crop,
                                     check_indices=False)
        self.ok = self.ok & (g["n_views"] == MAX_VIEW_NUM).all() & (g["status"] == 0).all()
        batch.crop_params.copy_(g["crop_params"].reshape(-1, 24))
        batch.intrinsics.copy_(g["intrinsics"].reshape(-1, 3, 3))
        batch.extrinsics.copy_(g["extrinsics"].reshape(-1, 4, 4))
        batch.src_index.copy_((self._src_base + g["cam_index"].long().clamp_(min=0)).reshape(-1).int())
        return batch


def make_batch(plan: dict, src_u8: torch.Tensor, device, independent_frames: bool = True) -> FrameBatch:
    """independent_frames: every hand-sample owns a temporal slot and starts without memory
    (`memory_idx=arange(S)`, `use_memory=False`, the throughput configuration of SURVEY.md section 8 d)."""
    dev = torch.device(device)
    s = plan["sample_range"].shape[0]
    sr = plan["sample_range"]
    return FrameBatch(
        src=src_u8.to(dev), cam_params=torch.from_numpy(plan["cam_params"]).to(dev),
        crop_params=torch.from_numpy(plan["crop_params"]).to(dev), src_index=torch.from_numpy(plan["src_index"]).to(dev),
        intrinsics=torch.from_numpy(plan["intrinsics"]).to(dev), extrinsics=torch.from_numpy(plan["extrinsics"]).to(dev),
        sample_range=torch.from_numpy(sr).to(dev),
        memory_idx=torch.arange(s, dtype=torch.long, device=dev) if independent_frames else torch.from_numpy(plan["hand_idx"]).to(dev),
        use_memory=torch.zeros(s, dtype=torch.uint8, device=dev), hand_idx=torch.from_numpy(plan["hand_idx"]).to(dev),
        n_slots=s if independent_frames else int(plan["hand_idx"].max()) + 1,
        all_multiview=bool(((sr[:, 1] - sr[:, 0]) == 2).all()))


# ----------------------------------------------------------------------------- the hot path
class HotPath:
    """warp -> backbone -> fuse/temporal/regress -> FK over one FrameBatch; all buffers preallocated.

    Inside `step` the engine's index checks run deferred (no stream synchronisation, so the host keeps launching ahead
    of the GPU) and its latency mode is off; both settings are put back at the end of the step, so the handle can be
    shared with a per-frame HandTracker.  A bad index tensor makes the device skip that call's work, and `check()` - call
    it wherever the records are consumed - raises IndexError for it."""

    def __init__(self, engine: _native.HipEngine, hand_model_mm: HandModel, known_skeleton: bool = True,
                 remap_mode: int = _native.UT_REMAP_CV2_FIXED, keep_crops: bool = False):
        """keep_crops: materialise the fp32 crop tensor (ut_warp_crops + ut_backbone, the crops are then in
        `self.crops`) instead of the fused ut_warp_backbone, whose crops stay u8 in the engine's workspace."""
        self.engine = engine
        self.keep_crops = keep_crops
        self.mode = _native.UT_MODE_KNOWN if known_skeleton else _native.UT_MODE_UNKNOWN
        self.remap_mode = remap_mode
        dev = engine.device
        self.hand_blob = torch.from_numpy(_native.hand_model_blob(
            hand_model_mm.joint_rotation_axes, hand_model_mm.joint_rest_positions, hand_model_mm.landmark_rest_positions,
            hand_model_mm.landmark_rest_bone_weights, hand_model_mm.landmark_rest_bone_indices)).reshape(1, 321).to(dev)
        self.skel = None
        if known_skeleton:   # mm -> m (lib/tracker/tracker.py:361-367)
            self.skel = torch.stack([hand_model_mm.joint_rotation_axes.float(),
                                     hand_model_mm.joint_rest_positions.float() * 0.001])[None].contiguous().to(dev)
        self._bufs = None

    def _buffers(self, b: FrameBatch):
        key = (b.n_crops, b.n_samples)
        if self._bufs is None or self._bufs[0] != key:
            dev = self.engine.device
            self._bufs = (key, torch.empty(b.n_crops if self.keep_crops else 0, arch.CROP, arch.CROP, device=dev),
                          torch.empty(b.n_crops, arch.FEAT_CH, arch.FEAT_HW, arch.FEAT_HW, device=dev),
                          torch.empty(b.n_samples, RECORD, device=dev))
            self.engine.reserve(b.n_crops, b.n_samples, b.n_slots)
        return self._bufs[1:]

    def step(self, b: FrameBatch) -> torch.Tensor:
        """[S,123] records (pose record | keypoints in mm)."""
        eng = self.engine
        crops, feat, rec = self._buffers(b)
        with eng.modes(deferred_checks=True, latency=False):
            if self.keep_crops:
                eng.warp_crops(b.src, b.cam_params, b.crop_params, b.src_index, self.remap_mode, out=crops)
                eng.backbone(crops, out=feat)
            else:
                eng.warp_backbone(b.src, b.cam_params, b.crop_params, b.src_index, self.remap_mode, out=feat)
            s = b.n_samples
            pose, _ = eng.fuse_temporal_regress(feat, b.intrinsics, b.extrinsics, b.sample_range, b.memory_idx,
                                                b.use_memory, b.hand_idx, b.n_slots, b.all_multiview, self.skel,
                                                self.mode, out=self._pose_buf(s))
            # FK consumes the pose records in place (row stride 60): metres -> mm, right hands un-mirrored
            kp = eng.fk(self.hand_blob, pose, pose[:, 22:], mirror=b.hand_idx, t_scale=1000.0,
                        ja_stride=arch.POSE_REC, xf_stride=arch.POSE_REC, n=s, out=self._kp_buf(s))
        rec[:, : arch.POSE_REC].copy_(pose)
        rec[:, arch.POSE_REC:].copy_(kp.reshape(s, -1))
        return rec

    def check(self):
        """Synchronises; raises IndexError if an index check failed in any step since the last call."""
        self.engine.poll_status()

    def _pose_buf(self, s):
        if getattr(self, "_pose", None) is None or self._pose.shape[0] != s:
            self._pose = torch.empty(s, arch.POSE_REC, device=self.engine.device)
        return self._pose

    def _kp_buf(self, s):
        if getattr(self, "_kp", None) is None or self._kp.shape[0] != s:
            self._kp = torch.empty(s, arch.N_LANDMARKS, 3, device=self.engine.device)
        return self._kp


# ----------------------------------------------------------------------------- sharding
def shard_frames(n_frames_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one frame."""
    base, extra = divmod(n_frames_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sequences(n_sequences_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Sequence mode (temporal memory engaged): the unit that shards is the SEQUENCE, not the frame.  Rank `rank` owns
    the contiguous block [lo, hi) of sequences for every time step, so a hand-sequence's temporal slot
    (`_mem_features[memory_idx]`, lib/models/temporal.py:101-137) lives on one rank for its whole life and no state ever
    crosses ranks; the records of a step are gathered in sequence order by `gather_records` like frame shards are.
    Mirrors the reference's only distribution idiom, `indices[rank::world_size]` (lib/data_utils/async_dataset.py:546-559),
    but contiguous to keep output order (SURVEY.md section 8 e)."""
    return shard_frames(n_sequences_total, rank, world)


def sequence_step_descriptors(hand_idx: torch.Tensor, first_step: bool) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """(memory_idx, use_memory, n_slots) of one time step of a rank's sequence block: hand-sample i of the step (row
    order of the crop plan: sequence-major, hands inside) keeps slot i on this rank at every step, and uses its memory
    from the second step on (`use_memory=False` only at step 0, run_inference_torch_data.py:39-85)."""
    s = int(hand_idx.shape[0])
    return (torch.arange(s, dtype=torch.long, device=hand_idx.device),
            torch.full((s,), 0 if first_step else 1, dtype=torch.uint8, device=hand_idx.device), s)


def gather_records(local: torch.Tensor, world: int, equal_counts: bool = False) -> torch.Tensor:
    """All-gather the per-rank [S_local, R] record blocks into the rank-ordered [sum S_local, R] on every rank
    (RCCL over xGMI when the process group is 'nccl'; 'gloo' in the CPU tests).

    Ranks may hold different numbers of records: `shard_frames` hands out blocks that differ by one frame, and a
    hand dropped by the confidence / visibility gate (lib/tracker/tracker.py:236-241, perspective_crop.py:168-178)
    shortens one rank's block.  So the counts are exchanged first (one 8-byte all-gather), every block is padded to the
    longest, gathered with ONE all_gather_into_tensor, and the padding is trimmed.  `equal_counts=True` skips the count
    exchange when the caller knows the blocks are equal (bench.py: equal frame blocks, two confident hands per frame)
    and lets a mismatch fail loudly in the collective's own size check instead of hanging."""
    if world == 1:
        return local
    import torch.distributed as dist
    local = local.contiguous()
    tail = tuple(local.shape[1:])
    if equal_counts:
        counts = [local.shape[0]] * world
    else:
        mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        allc = torch.empty(world, dtype=torch.int64, device=local.device)
        _all_gather_flat(allc, mine, world)
        counts = [int(c) for c in allc.tolist()]
    longest = max(counts)
    if local.shape[0] != longest:
        padded = local.new_zeros((longest,) + tail)
        padded[: local.shape[0]] = local
        local = padded
    out = torch.empty((world * longest,) + tail, dtype=local.dtype, device=local.device)
    _all_gather_flat(out, local, world)
    if all(c == longest for c in counts):
        return out
    return torch.cat([out[r * longest: r * longest + c] for r, c in enumerate(counts)], 0)


def _all_gather_flat(out: torch.Tensor, local: torch.Tensor, world: int) -> None:
    import torch.distributed as dist
    if dist.get_backend() == "gloo":       # gloo has no all_gather_into_tensor, and no all_gather of device tensors
        if local.device.type == "cuda":
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather(list(host.chunk(world, 0)), local.cpu())
            out.copy_(host)
        else:
            dist.all_gather(list(out.chunk(world, 0)), local)
    else:
        dist.all_gather_into_tensor(out, local)
