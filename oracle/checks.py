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


def _crop_from_row(row: np.ndarray) -> dict:
    """A packed crop-camera row of ut_warp_crops (fx fy cx cy | R(9) t(3)) as the oracle's camera dict."""
    t = np.eye(4)
    t[:3, :3] = np.asarray(row[4:13], np.float64).reshape(3, 3)
    t[:3, 3] = row[13:16]
    return {"w": arch.CROP, "h": arch.CROP, "f": (float(row[0]), float(row[1])), "c": (float(row[2]), float(row[3])),
            "k": None, "T": t}


def oracle_frames(sd_np, lab, hm_np, frame_ids, frames_u8: np.ndarray, known: bool = True, crops_override=None,
                  remap_mode: str = "cv2", plan: dict = None, network: bool = True):
    """The reference's per-frame path restated on the CPU for a list of label frames, batched at the
    network stage.  frames_u8 [F,4,H,W].  Returns dict(crops, pose60, keypoints_mm, hand_idx).
    plan: a product crop plan for the same frames (pipeline.crop_plan_from_labels): the oracle then resamples with THOSE
    crop cameras and feeds the network THEIR intrinsics / extrinsics instead of generating its own - the "identical inputs"
    form of the comparison at the camera level.  remap_mode "cv2" (OpenCV's 8-bit arithmetic: the reference) or "float"."""
    crops, intr, extr, ranges, hand_idx = [], [], [], [], []
    n_lab = lab["joint_angles"].shape[0]
    n_cams = lab["cameras"].shape[0]
    if plan is not None:
        for n, row in enumerate(plan["crop_params"]):
            fo, ci = divmod(int(plan["src_index"][n]), n_cams)
            cams = _oracle_cams(lab, int(frame_ids[fo]) % n_lab)
            img = ref_camera.warp_image(cams[ci], _crop_from_row(row), frames_u8[fo, ci], remap_mode)
            crops.append(img.astype(np.float32) / np.float32(255.0))
        intr, extr = list(plan["intrinsics"]), list(plan["extrinsics"])
        ranges, hand_idx = [tuple(r) for r in plan["sample_range"]], list(plan["hand_idx"])
    else:
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
                    img = ref_camera.warp_image(cams[ci], crop, frames_u8[fo, ci], remap_mode)
                    crops.append(img.astype(np.float32) / np.float32(255.0))
                    k, e = ref_camera.network_inputs_for_crop(crop)
                    intr.append(k)
                    extr.append(e)
                ranges.append((start, len(crops)))
                hand_idx.append(h)
    crops = np.stack(crops)
    hand_idx = np.asarray(hand_idx, np.int64)
    if not network:
        return {"crops": crops, "hand_idx": hand_idx}
    net_in = crops if crops_override is None else crops_override
    s = len(ranges)
    om = ref_model.OracleModel(sd_np)
    axes = torch.from_numpy(hm_np["joint_rotation_axes"].astype(np.float32))
    rest = torch.from_numpy((hm_np["joint_rest_positions"] * np.float32(0.001)).astype(np.float32))
    o = om.forward(torch.from_numpy(net_in), torch.from_numpy(np.stack(intr).astype(np.float32)),
                   torch.from_numpy(np.stack(extr).astype(np.float32)),
                   torch.tensor(ranges, dtype=torch.long), torch.arange(s), torch.zeros(s, dtype=torch.bool),
                   torch.from_numpy(hand_idx), axes, rest, known_skeleton=known)
    xf = o["wrist_xfs"].numpy().copy()
    xf_mm = xf.copy()
    xf_mm[:, :3, 3] *= 1000.0                                   # lib/tracker/tracker.py:379
    xf_mm[hand_idx == 1, :, 0] *= -1                             # lib/tracker/perspective_crop.py:48-49
    kp = ref_fk.skin_landmarks(hm_np, o["joint_angles"].numpy(), xf_mm)
    return {"crops": crops, "joint_angles": o["joint_angles"].numpy(), "wrist_xfs": xf, "keypoints_mm": kp,
            "hand_idx": hand_idx, "skel_scales": None if o["skel_scales"] is None else o["skel_scales"].numpy()}


def run_small_end_to_end(sd_np, n_frames: int = 2, device: str = "cuda:0", known: bool = True, conv: str = "fp32") -> Dict[str, float]:
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    hm_np = {k[3:]: v for k, v in lab.items() if k.startswith("hm.")}
    frame_ids = [int(i * 37) % 369 for i in range(n_frames)]
    frames = synth.synthetic_frames(n_frames, seed=5)
    eng = _native.HipEngine(sd_np, device)
    try:
        eng.set_conv_arithmetic(conv)        # "split_f16_always": the split-fp16 kernels also for this handful of crops
        plan = pipeline.crop_plan_from_labels(lab, hm, frame_ids)
        batch = pipeline.make_batch(plan, torch.from_numpy(frames.reshape(-1, 480, 636)), device)
        hot = pipeline.HotPath(eng, hm, known_skeleton=known, keep_crops=True)
        rec = hot.step(batch).cpu().numpy()
        hot.check()
        gpu_crops = hot._bufs[1].cpu().numpy()
    finally:
        eng.close()
    # identical inputs at the camera level: the oracle resamples with the product's crop cameras (and must reproduce its
    # crops bit for bit), then runs its own network on its own crops
    ref = oracle_frames(sd_np, lab, hm_np, frame_ids, frames, known, plan=plan)
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
    return {"seconds": dt, "hand_frames": int(out["keypoints_mm"].shape[0]), "oracle": out, "frames": frames, "frame_ids": frame_ids}


ANGLE_TOL_RAD, KEYPOINT_TOL_MM = 1e-4, 1e-3        # BASELINE.json north_star


def _gpu_records(sd_np, plan, frames_u8, device, conv, remap_mode):
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    eng = _native.HipEngine(sd_np, device)
    try:
        eng.set_conv_arithmetic("fp32" if conv == "fp32" else "split_f16_always")
        batch = pipeline.make_batch(plan, torch.from_numpy(frames_u8.reshape(-1, 480, 636)), device)
        hot = pipeline.HotPath(eng, hm, known_skeleton=True, keep_crops=True, remap_mode=remap_mode)
        rec = hot.step(batch).cpu().numpy()
        hot.check()
        return rec, hot._bufs[1].cpu().numpy()
    finally:
        eng.close()


def _errors(rec, ref):
    ang = np.abs(rec[:, :22] - ref["joint_angles"]).max(axis=1)
    kp = np.linalg.norm(rec[:, 60:].reshape(-1, 21, 3) - ref["keypoints_mm"], axis=-1).max(axis=1)
    return ang, kp


def batched_parity(sd_np, timed: Dict, device: str, convs=("fp32", "split_f16"),
                   legs=("identical_cameras", "own_cameras_cv2", "own_cameras_float")):
    """The batched hot path (pipeline.HotPath, one step over all frames of the CPU-baseline sample, raw images -> keypoints)
    against the oracle, once per backbone arithmetic in `convs` ("fp32" / "split_f16": the split kernels are forced for every
    launch size so that the whole sample goes through them; the oracle's passes are shared).  Returns one dict per
    arithmetic.  Three legs:

      identical_cameras   the oracle resamples with the PRODUCT's crop cameras (identical inputs at the camera level, the
                          parity contract of north_star): its crops must equal the GPU's bit for bit (the coordinate map is
                          pinned to the reference's by tests/test_gpu_parity.py::test_warp_coordinate_map_equals_reference_goldens,
                          the 8-bit remap arithmetic is integer), and every hand-frame is inside 1e-4 rad / 1e-3 mm.
      own_cameras_cv2     the oracle generates its OWN crop cameras (`timed` = time_oracle's result: the CPU baseline).  The two
                          sides' forward kinematics of the crop points (fp32: numpy there, fk.hip here) differ in the last
                          bit, the crop cameras by ~1e-7 relative, the maps by ~1e-5 px - and cv2's 8-bit remap quantises
                          coordinates to 1/32 px, so ~1e-4 of the pixels land on the other side of a rounding boundary and move
                          by a grey level.  Reported: how many hand-frames have identical crops, and how many are outside the
                          tolerance (the reference against itself on another BLAS would do the same).
      own_cameras_float   the same in UT_REMAP_FLOAT mode (continuous in the coordinates: a 1e-5 px shift moves a pixel by
                          ~1e-5 of its range, nothing flips): every hand-frame inside tolerance from raw images."""
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    hm_np = {k[3:]: v for k, v in lab.items() if k.startswith("hm.")}
    plan = pipeline.crop_plan_from_labels(lab, hm, timed["frame_ids"])
    sr = np.asarray(plan["sample_range"])
    t0 = time.perf_counter()
    ref_ident = oracle_frames(sd_np, lab, hm_np, timed["frame_ids"], timed["frames"], plan=plan) if "identical_cameras" in legs else None
    ref_float = oracle_frames(sd_np, lab, hm_np, timed["frame_ids"], timed["frames"], remap_mode="float") if "own_cameras_float" in legs else None
    oracle_s = time.perf_counter() - t0
    results = []
    for conv in convs:
        out: Dict[str, object] = {"conv_arithmetic": conv, "tolerance": "1e-4 rad / 1e-3 mm", "extra_oracle_cpu_seconds": round(oracle_s, 1)}
        rec, crops = _gpu_records(sd_np, plan, timed["frames"], device, conv, _native.UT_REMAP_CV2_FIXED)
        out["hand_frames"] = int(rec.shape[0])
        if ref_ident is not None:
            ang, kp = _errors(rec, ref_ident)
            out["identical_cameras"] = {
                "crop_pixels_differing": int((crops != ref_ident["crops"]).sum()),
                "max_joint_angle_err_rad": float(ang.max()), "max_keypoint_err_mm": float(kp.max()),
                "hand_frames_outside_tolerance": int(((ang >= ANGLE_TOL_RAD) | (kp >= KEYPOINT_TOL_MM)).sum())}
        if "own_cameras_cv2" in legs:
            ref = timed["oracle"]
            same = np.array([np.array_equal(crops[a:b], ref["crops"][a:b]) for a, b in sr])
            ang, kp = _errors(rec, ref)
            out["own_cameras_cv2"] = {
                "hand_frames_with_identical_crops": int(same.sum()),
                "crop_pixel_mismatch_fraction": float((crops != ref["crops"]).mean()),
                "max_joint_angle_err_rad": float(ang.max()), "max_keypoint_err_mm": float(kp.max()),
                "hand_frames_outside_tolerance": int(((ang >= ANGLE_TOL_RAD) | (kp >= KEYPOINT_TOL_MM)).sum())}
        if ref_float is not None:
            rec_f, crops_f = _gpu_records(sd_np, plan, timed["frames"], device, conv, _native.UT_REMAP_FLOAT)
            ang, kp = _errors(rec_f, ref_float)
            out["own_cameras_float"] = {
                "max_crop_abs_diff": float(np.abs(crops_f - ref_float["crops"]).max()),
                "max_joint_angle_err_rad": float(ang.max()), "max_keypoint_err_mm": float(kp.max()),
                "hand_frames_outside_tolerance": int(((ang >= ANGLE_TOL_RAD) | (kp >= KEYPOINT_TOL_MM)).sum())}
        results.append(out)
    return results


# ----------------------------------------------------------------------------- recording_00 as a sequence
def _gt_tracking(lab, fi):
    from absolutetrack_amd.tracker import SingleHandPose
    return {h: SingleHandPose(joint_angles=lab["joint_angles"][fi, h], wrist_xform=lab["wrist_transforms"][fi, h],
                              hand_confidence=float(lab["hand_confidences"][fi, h])) for h in (0, 1)}


def _input_frame(lab, fi, frame_u8):
    from absolutetrack_amd.tracker import InputFrame, ViewData
    cams = pipeline.cameras_for_frame(lab, fi)
    return InputFrame(views=[ViewData(image=frame_u8[ci], camera=cams[ci], camera_angle=lab["camera_angles"][ci])
                             for ci in range(len(cams))]), cams


def _oracle_keypoints(hm_np, ja, xf_m, hand_idx):
    """tracker.py:379 (m -> mm) + perspective_crop.py:48-49 (right hands mirrored) + FK, on the oracle's outputs."""
    xf = np.array(xf_m, np.float32)
    xf[:3, 3] *= np.float32(1000.0)
    if hand_idx == 1:
        xf[:, 0] *= -1
    return ref_fk.skin_landmarks(hm_np, np.asarray(ja, np.float32), xf)


def _scaled_np(hm_np, s):
    out = dict(hm_np)
    out["joint_rest_positions"] = (hm_np["joint_rest_positions"].astype(np.float32) * np.float32(s)).astype(np.float32)
    out["landmark_rest_positions"] = (hm_np["landmark_rest_positions"].astype(np.float32) * np.float32(s)).astype(np.float32)
    return out


def run_recording00(sd_np, device: str = "cuda:0", known: bool = True, n_frames: int = 0, n_images: int = 8,
                    n_calibration_samples: int = 30) -> Dict[str, float]:
    """All label frames of sample_data/recording_00 as ONE sequence through the product's drop-in HandTracker, the
    way the reference's eval scripts drive it, next to the oracle fed the product's crops frame by frame (each side
    keeps its own temporal memory, validity history and - in the unknown-skeleton flow - its own calibrated scale):

      known   run_eval_known_skeleton.py:68-93    gen_crop_cameras(min_num_crops=1) -> track_frame -> FK -> MPJPE
      unknown run_eval_unknown_skeleton.py:49-78  pass 1: track_frame_and_calibrate_scale until 30 scale samples,
                                        :99-126   reset_history, pass 2 = the known flow on the calibrated generic model

    Images are synthetic (the mp4 is a missing blob): n_images seeded u8 frames, frame i uses image i mod n_images.
    Returns the MPJPE of both sides against the label keypoints, their difference, and the worst joint-angle /
    wrist / keypoint deviation between the two over every hand-frame.  Also returns the oracle's CPU seconds."""
    from lib.common.hand import scaled_hand_model
    from lib.models.umetrack_model import UmeTrackModel
    from lib.tracker.perspective_crop import landmarks_from_hand_pose
    from lib.tracker.tracker import HandTracker, HandTrackerOpts
    lab = pipeline.load_labels()
    gt_hm = pipeline.hand_model_from_labels(lab)
    gt_hm_np = {k[3:]: v for k, v in lab.items() if k.startswith("hm.")}
    n = int(n_frames) if n_frames else lab["joint_angles"].shape[0]
    images = synth.synthetic_frames(n_images, seed=21)
    model = UmeTrackModel(sd_np)
    model.eval()
    trk = HandTracker(model, HandTrackerOpts())
    om = ref_model.OracleModel(sd_np)
    angles = list(lab["camera_angles"])
    cpu_s = 0.0
    res = {"mode": "known" if known else "unknown", "frames": n}

    def oracle_step(fd, desc, skel, known_flow):
        nonlocal cpu_s
        args = [fd.left_images.cpu(), fd.intrinsics.cpu(), fd.extrinsics_xf.cpu(), desc.sample_range.cpu(),
                desc.memory_idx.cpu(), desc.use_memory.cpu(), desc.hand_idx.cpu()]
        t0 = time.perf_counter()
        if known_flow:
            o = om.forward(*args, skel.joint_rotation_axes.cpu(), skel.joint_rest_positions.cpu(), True)
        else:
            o = om.forward(*args, known_skeleton=False)
        cpu_s += time.perf_counter() - t0
        return o

    track_hm, track_hm_np = gt_hm, gt_hm_np
    if not known:
        g = dict(np.load(pipeline._DATA.replace("recording_00_labels", "generic_hand_model")))
        from absolutetrack_amd.hand import HandModel
        generic = HandModel(**{k: torch.from_numpy(np.asarray(v)) for k, v in g.items()})
        scales_p, scales_o = [], []
        for fi in range(n):
            sample, cams = _input_frame(lab, fi, images[fi % n_images])
            cc = trk.gen_crop_cameras(cams, angles, gt_hm, _gt_tracking(lab, fi), min_num_crops=2)
            if cc:
                fd, desc, _ = trk._make_inputs(sample, None, cc)
                o = oracle_step(fd, desc, None, False)
                scales_o += [float(v) for v in o["skel_scales"]]
            r = trk.track_frame_and_calibrate_scale(sample, cc)
            scales_p += [float(r.predicted_scales[h]) for h in r.hand_poses.keys()]
            if n_calibration_samples and len(scales_p) >= n_calibration_samples:
                scales_p, scales_o = scales_p[:n_calibration_samples], scales_o[:n_calibration_samples]
                break
        assert len(scales_p) == len(scales_o) > 0
        mean_p, mean_o = float(np.mean(scales_p)), float(np.mean(scales_o))
        res.update(calibration_samples=len(scales_p), scale_mean_build=mean_p, scale_mean_oracle=mean_o,
                   scale_mean_abs_diff=abs(mean_p - mean_o),
                   scale_max_abs_diff=float(np.abs(np.array(scales_p) - np.array(scales_o)).max()))
        track_hm = scaled_hand_model(generic, mean_p)
        track_hm_np = _scaled_np({k: np.asarray(v) for k, v in g.items()}, mean_o)
        trk.reset_history()

    gt_kp = np.zeros((2, n, 21, 3))
    kp_p, kp_o = np.zeros_like(gt_kp), np.zeros_like(gt_kp)
    valid = np.zeros((2, n), bool)
    d_ja = d_t = 0.0
    o_valid = np.zeros(2, bool)
    for fi in range(n):
        sample, cams = _input_frame(lab, fi, images[fi % n_images])
        gt = _gt_tracking(lab, fi)
        cc = trk.gen_crop_cameras(cams, angles, gt_hm, gt, min_num_crops=1)
        if not cc:
            trk.track_frame(sample, track_hm, cc)
            o_valid[:] = False
            continue
        fd, desc, skel = trk._make_inputs(sample, track_hm, cc)
        assert desc.use_memory.cpu().numpy().tolist() == o_valid[desc.hand_idx.cpu().numpy()].tolist()
        if not known:   # the oracle regresses with ITS calibrated skeleton (metres, tracker.py:361-367)
            skel = type(skel)(joint_rotation_axes=skel.joint_rotation_axes,
                              joint_rest_positions=torch.from_numpy(track_hm_np["joint_rest_positions"] * np.float32(0.001)))
        o = oracle_step(fd, desc, skel, True)
        r = trk.track_frame(sample, track_hm, cc)
        hands = desc.hand_idx.cpu().numpy().tolist()
        for i, h in enumerate(hands):
            pose = r.hand_poses[h]
            kp_p[h, fi] = landmarks_from_hand_pose(track_hm, pose, h)
            gt_kp[h, fi] = landmarks_from_hand_pose(gt_hm, gt[h], h)
            kp_o[h, fi] = _oracle_keypoints(track_hm_np, o["joint_angles"][i].numpy(), o["wrist_xfs"][i].numpy(), h)
            valid[h, fi] = True
            d_ja = max(d_ja, float(np.abs(pose.joint_angles - o["joint_angles"][i].numpy()).max()))
            d_t = max(d_t, float(np.abs(pose.wrist_xform[:3, 3] - o["wrist_xfs"][i].numpy()[:3, 3] * 1000.0).max()))
        o_valid[:] = [h in hands for h in (0, 1)]
    model._drop_engine()
    mpjpe = lambda kp: float(np.linalg.norm((gt_kp - kp)[valid], axis=-1).mean(axis=-1).mean())
    res.update(hand_frames=int(valid.sum()), mpjpe_build_mm=mpjpe(kp_p), mpjpe_oracle_mm=mpjpe(kp_o),
               max_joint_angle_err_rad=d_ja, max_wrist_translation_err_mm=d_t,
               max_keypoint_err_mm=float(np.abs(kp_p - kp_o)[valid].max()), oracle_cpu_seconds=cpu_s)
    res["mpjpe_delta_mm"] = abs(res["mpjpe_build_mm"] - res["mpjpe_oracle_mm"])
    return res
