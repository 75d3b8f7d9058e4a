"""Seeded input scenarios shared by the golden generator and the parity tests
(TEST INFRASTRUCTURE ONLY).  Everything is regenerated from the counter-based RNG of
absolutetrack_amd.synth, so the fixtures under tests/golden/ only hold OUTPUTS.

The model scenarios exercise, per SURVEY.md section 8(c) G2: a mixed 1-view/2-view
batch, three consecutive temporal steps with use_memory F->T->T, a slot drop,
permuted slots with state growth, both hand indices and both regress modes.
"""
import os
from typing import Dict, List

import numpy as np

from absolutetrack_amd import synth

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "absolutetrack_amd", "data", "recording_00_labels.npz")


def labels() -> Dict[str, np.ndarray]:
    return dict(np.load(_DATA))


def hand_model_mm() -> Dict[str, np.ndarray]:
    lab = labels()
    return {k[3:]: v for k, v in lab.items() if k.startswith("hm.")}


def skeleton_m():
    """(axes [22,3], rest [22,3] in metres) as HandTracker._make_inputs builds them
    (lib/tracker/tracker.py:361-367: scaled_hand_model(hand_model_mm, 0.001))."""
    hm = hand_model_mm()
    return (hm["joint_rotation_axes"].astype(np.float32),
            (hm["joint_rest_positions"] * np.float32(0.001)).astype(np.float32))


def _rigid(key: str, n: int, seed: int) -> np.ndarray:
    """n world->eye extrinsics: random rotation (axis-angle, |angle| < 1.2 rad), t ~ U(-0.3,0.3) m."""
    u = synth.counter_uniform(key, n * 6, seed).reshape(n, 6) * 2 - 1
    out = np.zeros((n, 4, 4), np.float64)
    for i in range(n):
        v = u[i, :3] * 1.2 / np.sqrt(3)
        th = np.linalg.norm(v)
        k = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        r = np.eye(3) + np.sin(th) / th * k + (1 - np.cos(th)) / th ** 2 * (k @ k)
        out[i, :3, :3] = r
        out[i, :3, 3] = u[i, 3:] * 0.3
        out[i, 3, 3] = 1
    return out.astype(np.float32)


def _intrinsics(key: str, n: int, seed: int) -> np.ndarray:
    f = 100 + 60 * synth.counter_uniform(key, n, seed)
    k = np.zeros((n, 3, 3), np.float32)
    k[:, 0, 0] = k[:, 1, 1] = f
    k[:, 0, 2] = k[:, 1, 2] = 47.5
    k[:, 2, 2] = 1
    return k


def _step(tag: str, seed: int, sample_range, memory_idx, use_memory, hand_idx) -> dict:
    sample_range = np.asarray(sample_range, np.int64)
    n = int(sample_range[-1, 1])
    return {"images": synth.synthetic_crops(n, seed=100 + seed),
            "intrinsics": _intrinsics(tag + ".K", n, seed),
            "extrinsics": _rigid(tag + ".X", n, seed),
            "sample_range": sample_range,
            "memory_idx": np.asarray(memory_idx, np.int64),
            "use_memory": np.asarray(use_memory, bool),
            "hand_idx": np.asarray(hand_idx, np.int64)}


def model_steps(known: bool) -> List[dict]:
    if known:
        return [
            _step("k", 0, [[0, 2], [2, 3], [3, 5]], [0, 1, 2], [False, False, False], [0, 1, 1]),
            _step("k", 1, [[0, 2], [2, 3], [3, 5]], [0, 1, 2], [True, True, True], [0, 1, 1]),
            _step("k", 2, [[0, 1], [1, 3]], [1, 2], [True, False], [1, 0]),            # slot 0 dropped
            _step("k", 3, [[0, 2], [2, 4], [4, 6], [6, 8]], [3, 0, 1, 2], [False, True, True, True], [0, 1, 0, 1]),
        ]
    return [
        _step("u", 0, [[0, 2], [2, 4]], [0, 1], [False, False], [0, 1]),
        _step("u", 1, [[0, 2], [2, 4], [4, 6], [6, 8]], [3, 0, 1, 2], [False, True, True, False], [1, 0, 1, 0]),
        _step("u", 2, [[0, 2], [2, 4]], [2, 0], [True, True], [0, 0]),
    ]


def torch_data_case(hand: int, seed: int = 0, n_frames: int = 4, h: int = 240, w: int = 320) -> Dict[str, np.ndarray]:
    """Inputs of _perspective_crop_images (lib/batched_dataset/data_transform.py:215-283) for one sequence, in the
    units the reference holds them after RawSample.scaled(0.001) (metres): two pinhole views per frame placed at the
    positions of fisheye cameras 1 and 2 of recording_00 and aimed near the hand (seeded offsets, the last frame far
    enough off that part of the crop leaves the source image), u8 frames, 63 enclosing points from the label pose."""
    from . import ref_camera
    lab = labels()
    hm = hand_model_mm()
    frames = [5 + 40 * i for i in range(n_frames)]
    lim = hm["joint_limits"].astype(np.float32)
    img = synth.synthetic_frames(n_frames, n_cams=2, h=h, w=w, seed=40 + seed + hand)
    ext = np.zeros((n_frames, 2, 4, 4), np.float32)
    intr = np.zeros((n_frames, 2, 3, 3), np.float32)
    pts = np.zeros((n_frames, 63, 3), np.float32)
    u = synth.counter_uniform(f"torch_data.{hand}", n_frames * 2 * 6, seed).reshape(n_frames, 2, 6)
    for i, fi in enumerate(frames):
        ja, wx = lab["joint_angles"][fi, hand], lab["wrist_transforms"][fi, hand]
        poses = (ja, lim[:, 0] * np.float32(0.5) + lim[:, 1] * np.float32(0.5), np.zeros(22, np.float32))
        p_mm = np.concatenate([ref_camera.landmarks_from_pose(hm, p, wx, hand) for p in poses], 0)
        pts[i] = (p_mm * 0.001).astype(np.float32)
        centre = (pts[i].min(0) + pts[i].max(0)).astype(np.float64) / 2
        for v, ci in enumerate((1, 2)):
            c2w = lab["camera_to_world_transforms"][fi, ci].copy()
            c2w[:3, 3] *= 0.001
            off = (u[i, v, :3] * 2 - 1) * (0.16 if i == n_frames - 1 else 0.05)
            ext[i, v] = ref_camera.look_at(np.linalg.inv(c2w), centre + off, 0.0).astype(np.float32)
            f = w * (0.7 + 0.2 * u[i, v, 3])
            intr[i, v] = [[f, 0, (w - 1) / 2 + 6 * (u[i, v, 4] - 0.5)], [0, f, (h - 1) / 2 + 6 * (u[i, v, 5] - 0.5)], [0, 0, 1]]
    return {"images": img, "extrinsics": ext, "intrinsics": intr, "crop_points": pts, "hand": hand}


def metrics_case(seed: int = 0, n_frames: int = 60) -> Dict[str, np.ndarray]:
    """Eval-result-shaped arrays (run_eval_known_skeleton.py:62-64): float32-valued keypoints in float64 arrays
    [2, T, 21, 3] (mm) - a smooth random walk as ground truth, tracked = gt + noise with a few gross errors so the
    PCK curve is not saturated - and a validity mask with gaps."""
    steps = synth.counter_normal("metrics.walk", 2 * n_frames * 63, seed).reshape(2, n_frames, 21, 3)
    gt = (np.cumsum(np.cumsum(steps * 0.4, axis=1), axis=1) + 300.0).astype(np.float32)
    noise = synth.counter_normal("metrics.noise", 2 * n_frames * 63, seed).reshape(2, n_frames, 21, 3) * 6.0
    gross = synth.counter_uniform("metrics.gross", 2 * n_frames, seed).reshape(2, n_frames, 1, 1) > 0.9
    tracked = (gt + noise + gross * 80.0).astype(np.float32)
    valid = synth.counter_uniform("metrics.valid", 2 * n_frames, seed).reshape(2, n_frames) > 0.2
    tracked[~valid] = 0
    gt_masked = gt.copy()
    gt_masked[~valid] = 0        # the eval scripts leave untracked frames at zero in both arrays
    return {"gt_keypoints": gt_masked.astype(np.float64), "tracked_keypoints": tracked.astype(np.float64),
            "valid_tracking": valid}


def idxbin_case(seed: int = 0) -> Dict[str, object]:
    """Small torch_data-style frames: a uniform uint8 image block per sequence, msgpack label dicts, and a ragged
    float32 file (lib/data_utils/idxbinfile.py supports frames of differing shape)."""
    u = synth.counter_uniform("idxbin.mono", 3 * 2 * 2 * 16 * 24, seed)
    mono = np.floor(u * 256).astype(np.uint8).reshape(3, 2, 2, 16, 24)
    labels = []
    for i in range(3):
        v = synth.counter_uniform(f"idxbin.lab{i}", 40, seed)
        labels.append({"hand": [float(i % 2)] * 2, "joint_angles": v[:8].reshape(2, 4).tolist(),
                       "extrinsics": v[8:40].reshape(2, 4, 4).tolist(),
                       "hand_model": {"joint_rest_positions": v[:6].reshape(2, 3).tolist(), "hand_scale": 1.0}})
    ragged = [synth.counter_normal(f"idxbin.r{i}", n, seed).astype(np.float32).reshape(shape)
              for i, (n, shape) in enumerate([(6, (2, 3)), (4, (4,)), (24, (2, 3, 4))])]
    return {"mono": mono, "labels": labels, "ragged": ragged}
