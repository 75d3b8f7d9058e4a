"""End-to-end checker (TEST INFRASTRUCTURE ONLY): runs the product's batched HIP path on a few
label frames and compares every stage with the CPU oracle.  Used by __graft_entry__.smoke(), the
GPU tests and bench.py's cpu_baseline leg - never by the product."""
import time
from typing import Dict

import numpy as np
import torch

from absolutetrack_amd import _native, arch, pipeline, synth
from oracle import ref_camera, ref_fk, ref_model


def _oracle_cams(lab, frame):
    names = pipeline._CAM_FIELDS
    return [ref_camera.camera_from_json(dict(zip(names, lab["cameras"][ci])) | {"DistortionModel": "FishEye62"},
                                        lab["camera_to_world_transforms"][frame, ci]) for ci in range(4)]


def oracle_frames(sd_np, lab, hm_np, frame_ids, frames_u8: np.ndarray, known: bool = True, crops_override=None):
    """The reference's per-frame path restated on the CPU for a list of label frames, batched at the
    network stage.  frames_u8 [F,4,H,W].  Returns dict(crops, pose60, keypoints_mm, hand_idx)."""
    crops, intr, extr, ranges, hand_idx = [], [], [], [], []
    n_lab = lab["joint_angles"].shape[0]
    for fo, f in enumerate(frame_ids):
        lf = int(f) % n_lab
        cams = _oracle_cams(lab, lf)
        for h in (0, 1):
            cc = ref_camera.gen_crop_cameras(cams, lab["camera_angles"], hm_np, lab["joint_angles"][lf, h],
                                             lab["wrist_transforms"][lf, h], h)
            if not cc:
                continue
            start = len(crops)
            for ci, crop in cc.items():
                img = ref_camera.warp_image(cams[ci], crop, frames_u8[fo, ci], "cv2")
                crops.append(img.astype(np.float32) / np.float32(255.0))
                k, e = ref_camera.network_inputs_for_crop(crop)
                intr.append(k)
                extr.append(e)
            ranges.append((start, len(crops)))
            hand_idx.append(h)
    crops = np.stack(crops)
    net_in = crops if crops_override is None else crops_override
    s = len(ranges)
    hand_idx = np.asarray(hand_idx, np.int64)
    om = ref_model.OracleModel(sd_np)
    axes = torch.from_numpy(hm_np["joint_rotation_axes"].astype(np.float32))
    rest = torch.from_numpy((hm_np["joint_rest_positions"] * np.float32(0.001)).astype(np.float32))
    o = om.forward(torch.from_numpy(net_in), torch.from_numpy(np.stack(intr)), torch.from_numpy(np.stack(extr)),
                   torch.tensor(ranges, dtype=torch.long), torch.arange(s), torch.zeros(s, dtype=torch.bool),
                   torch.from_numpy(hand_idx), axes, rest, known_skeleton=known)
    xf = o["wrist_xfs"].numpy().copy()
    xf_mm = xf.copy()
    xf_mm[:, :3, 3] *= 1000.0                                   # lib/tracker/tracker.py:379
    xf_mm[hand_idx == 1, :, 0] *= -1                             # lib/tracker/perspective_crop.py:48-49
    kp = ref_fk.skin_landmarks(hm_np, o["joint_angles"].numpy(), xf_mm)
    return {"crops": crops, "joint_angles": o["joint_angles"].numpy(), "wrist_xfs": xf, "keypoints_mm": kp,
            "hand_idx": hand_idx, "skel_scales": None if o["skel_scales"] is None else o["skel_scales"].numpy()}


def run_small_end_to_end(sd_np, n_frames: int = 2, device: str = "cuda:0", known: bool = True) -> Dict[str, float]:
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    hm_np = {k[3:]: v for k, v in lab.items() if k.startswith("hm.")}
    frame_ids = [int(i * 37) % 369 for i in range(n_frames)]
    frames = synth.synthetic_frames(n_frames, seed=5)
    eng = _native.HipEngine(sd_np, device)
    try:
        plan = pipeline.crop_plan_from_labels(lab, hm, frame_ids)
        batch = pipeline.make_batch(plan, torch.from_numpy(frames.reshape(-1, 480, 636)), device)
        hot = pipeline.HotPath(eng, hm, known_skeleton=known)
        rec = hot.step(batch).cpu().numpy()
        gpu_crops = hot._bufs[1].cpu().numpy()
    finally:
        eng.close()
    ref = oracle_frames(sd_np, lab, hm_np, frame_ids, frames, known, crops_override=gpu_crops)
    crop_diff = np.abs(gpu_crops - ref["crops"])
    return {
        "hand_frames": int(rec.shape[0]),
        "crop_mismatch_fraction": float((crop_diff > 0).mean()),
        "crop_max_abs_diff": float(crop_diff.max()),
        "max_joint_angle_err_rad": float(np.abs(rec[:, :22] - ref["joint_angles"]).max()),
        "max_wrist_translation_err_mm": float(np.abs(rec[:, 22:38].reshape(-1, 4, 4)[:, :3, 3] - ref["wrist_xfs"][:, :3, 3]).max() * 1000),
        "max_keypoint_err_mm": float(np.abs(rec[:, 60:].reshape(-1, 21, 3) - ref["keypoints_mm"]).max()),
    }


def time_oracle(sd_np, n_frames: int, threads: int) -> Dict[str, float]:
    """CPU baseline: the oracle's full per-frame path (resample + network + FK) on `n_frames` label frames."""
    torch.set_num_threads(threads)
    lab = pipeline.load_labels()
    hm_np = {k[3:]: v for k, v in lab.items() if k.startswith("hm.")}
    frames = synth.synthetic_frames(min(n_frames, 8), seed=5)
    frames = np.concatenate([frames] * ((n_frames + frames.shape[0] - 1) // frames.shape[0]))[:n_frames]
    frame_ids = list(range(n_frames))
    t0 = time.perf_counter()
    out = oracle_frames(sd_np, lab, hm_np, frame_ids, frames)
    dt = time.perf_counter() - t0
    return {"seconds": dt, "hand_frames": int(out["keypoints_mm"].shape[0])}
