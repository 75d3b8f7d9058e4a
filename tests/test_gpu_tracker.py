"""GPU tests of the host-facing call surface (HandTracker / UmeTrackModel / skin_landmarks through the `lib.*`
module paths the reference's scripts import) and size-independent properties at BASELINE.json's batch sizes."""
import os

import numpy as np
import pytest
import torch

from absolutetrack_amd import _native, arch, pipeline, synth
from oracle import checks, ref_camera, ref_fk, ref_model

pytestmark = pytest.mark.gpu
ROOT_DIR = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def labels():
    return pipeline.load_labels()


@pytest.fixture(scope="module")
def hand_model(labels):
    return pipeline.hand_model_from_labels(labels)


def _oracle_hm(labels):
    return {k[3:]: v for k, v in labels.items() if k.startswith("hm.")}


def _input_frame(labels, fi, frames_u8):
    from lib.tracker.tracker import InputFrame, ViewData
    cams = pipeline.cameras_for_frame(labels, fi)
    return InputFrame(views=[ViewData(image=frames_u8[ci], camera=cams[ci], camera_angle=labels["camera_angles"][ci])
                             for ci in range(4)]), cams


def _gt(labels, fi):
    from lib.tracker.tracking_result import SingleHandPose
    return {h: SingleHandPose(joint_angles=labels["joint_angles"][fi, h], wrist_xform=labels["wrist_transforms"][fi, h],
                              hand_confidence=labels["hand_confidences"][fi, h]) for h in (0, 1)}


def test_reference_script_flow_matches_oracle(labels, hand_model):
    """The loop body of run_eval_known_skeleton.py:68-89 through the drop-in `lib` modules, three consecutive
    frames (temporal memory engaged from the second), against the oracle fed with the same crops."""
    from lib.models.umetrack_model import UmeTrackModel
    from lib.tracker.perspective_crop import landmarks_from_hand_pose
    from lib.tracker.tracker import HandTracker, HandTrackerOpts
    sd = synth.synthetic_state_dict(0)
    model = UmeTrackModel(sd)
    model.eval()
    trk = HandTracker(model, HandTrackerOpts())
    om = ref_model.OracleModel(sd)
    hm_np = _oracle_hm(labels)
    frames = synth.synthetic_frames(3, seed=9)
    valid = np.zeros(2, bool)
    for step, fi in enumerate((10, 11, 12)):
        sample, cams = _input_frame(labels, fi, frames[step])
        gt = _gt(labels, fi)
        crop_cameras = trk.gen_crop_cameras(cams, list(labels["camera_angles"]), hand_model, gt, min_num_crops=1)
        assert sorted(crop_cameras) == [0, 1] and all(len(v) == 2 for v in crop_cameras.values())
        # oracle crop cameras agree with the product's
        ocams = checks._oracle_cams(labels, fi)
        for h in (0, 1):
            oc = ref_camera.gen_crop_cameras(ocams, labels["camera_angles"], hm_np, labels["joint_angles"][fi, h],
                                             labels["wrist_transforms"][fi, h], h)
            assert list(oc) == list(crop_cameras[h])
            for ci in oc:
                np.testing.assert_allclose(oc[ci]["T"], crop_cameras[h][ci].camera_to_world_xf, atol=1e-6)
                np.testing.assert_allclose(oc[ci]["f"], crop_cameras[h][ci].f, rtol=1e-6)
        fd, desc, skel = trk._make_inputs(sample, hand_model, crop_cameras)
        assert fd.left_images.shape == (4, 96, 96) and desc.use_memory.tolist() == valid.tolist()
        res = trk.track_frame(sample, hand_model, crop_cameras)
        o = om.forward(fd.left_images.cpu(), fd.intrinsics.cpu(), fd.extrinsics_xf.cpu(), desc.sample_range.cpu(),
                       desc.memory_idx.cpu(), desc.use_memory.cpu(), desc.hand_idx.cpu(),
                       skel.joint_rotation_axes.cpu(), skel.joint_rest_positions.cpu(), True)
        assert sorted(res.hand_poses) == [0, 1] and res.num_views == {0: 2, 1: 2} and res.predicted_scales == {}
        for i, h in enumerate(desc.hand_idx.tolist()):
            pose = res.hand_poses[h]
            assert np.abs(pose.joint_angles - o["joint_angles"][i].numpy()).max() < 1e-4
            want_xf = o["wrist_xfs"][i].numpy().copy()
            want_xf[:3, 3] *= 1000.0
            assert np.abs(pose.wrist_xform[:3, 3] - want_xf[:3, 3]).max() < 1e-3           # mm
            kp = landmarks_from_hand_pose(hand_model, pose, h)
            xf = want_xf.copy()
            if h == 1:
                xf[:, 0] *= -1
            want_kp = ref_fk.skin_landmarks(hm_np, o["joint_angles"][i].numpy(), xf)
            assert kp.shape == (21, 3) and np.abs(kp - want_kp).max() < 1e-3              # mm
        valid[:] = True
    # no hands -> empty result and history reset (lib/tracker/tracker.py:268-271)
    assert trk.track_frame(sample, hand_model, {}).hand_poses == {}
    assert not trk._valid_tracking_history.any()


def test_staged_per_frame_path_equals_the_general_path(labels, hand_model):
    """HandTracker.track_frame normally runs staged (one upload of images + parameter rows, the fused resample+backbone
    entry, FK of the regressed poses in the same launch sequence, one read-back, landmarks remembered for
    landmarks_from_hand_pose).  It must return exactly what the general tensor-by-tensor path returns, frame after
    frame with the temporal memory engaged, and the remembered landmarks must be exactly the FK of the returned pose."""
    from lib.models.umetrack_model import UmeTrackModel
    from lib.tracker.perspective_crop import landmarks_from_hand_pose
    from lib.tracker.tracker import HandTracker, HandTrackerOpts
    from absolutetrack_amd import tracker as tk
    sd = synth.synthetic_state_dict(0)
    fast = HandTracker(UmeTrackModel(sd), HandTrackerOpts())
    slow = HandTracker(UmeTrackModel(sd), HandTrackerOpts())
    slow._run_staged = lambda *a, **k: None            # force the general path
    frames = synth.synthetic_frames(4, seed=13)
    for step, fi in enumerate((100, 101, 102, 103)):
        sample, cams = _input_frame(labels, fi, frames[step])
        gt = _gt(labels, fi)
        if step == 2:
            gt = {1: gt[1]}                             # one hand only: other shapes, a slot drops out
        cc = fast.gen_crop_cameras(cams, list(labels["camera_angles"]), hand_model, gt, min_num_crops=1)
        a = fast.track_frame(sample, hand_model, cc)
        b = slow.track_frame(sample, hand_model, cc)
        assert sorted(a.hand_poses) == sorted(b.hand_poses) == sorted(gt) and a.num_views == b.num_views
        for h in a.hand_poses:
            assert np.array_equal(a.hand_poses[h].joint_angles, b.hand_poses[h].joint_angles)
            assert np.array_equal(a.hand_poses[h].wrist_xform, b.hand_poses[h].wrist_xform)
            tk._landmark_memo.items.clear()
            direct = landmarks_from_hand_pose(hand_model, b.hand_poses[h], h)       # FK launch
        assert (fast._valid_tracking_history == slow._valid_tracking_history).all()
    # remembered landmarks: track_frame's and gen_crop_cameras' (label poses) are the FK results, bit for bit
    sample, cams = _input_frame(labels, 104, frames[0])
    gt = _gt(labels, 104)
    cc = fast.gen_crop_cameras(cams, list(labels["camera_angles"]), hand_model, gt, min_num_crops=1)
    res = fast.track_frame(sample, hand_model, cc)
    for h in (0, 1):
        for pose in (res.hand_poses[h], gt[h]):
            assert tk._landmark_memo.get(hand_model, h, pose.joint_angles, pose.wrist_xform) is not None
            remembered = landmarks_from_hand_pose(hand_model, pose, h)
            keep = list(tk._landmark_memo.items)
            tk._landmark_memo.items.clear()
            assert np.array_equal(remembered, landmarks_from_hand_pose(hand_model, pose, h))
            tk._landmark_memo.items[:] = keep


def test_calibration_path_matches_oracle(labels, hand_model):
    from lib.models.umetrack_model import UmeTrackModel
    from lib.tracker.tracker import HandTracker, HandTrackerOpts
    sd = synth.synthetic_state_dict(0)
    trk = HandTracker(UmeTrackModel(sd), HandTrackerOpts())
    om = ref_model.OracleModel(sd)
    frames = synth.synthetic_frames(1, seed=3)
    sample, cams = _input_frame(labels, 50, frames[0])
    crop_cameras = trk.gen_crop_cameras(cams, list(labels["camera_angles"]), hand_model, _gt(labels, 50), min_num_crops=2)
    fd, desc, _ = trk._make_inputs(sample, None, crop_cameras)
    res = trk.track_frame_and_calibrate_scale(sample, crop_cameras)
    o = om.forward(fd.left_images.cpu(), fd.intrinsics.cpu(), fd.extrinsics_xf.cpu(), desc.sample_range.cpu(),
                   desc.memory_idx.cpu(), desc.use_memory.cpu(), desc.hand_idx.cpu(), known_skeleton=False)
    for i, h in enumerate(desc.hand_idx.tolist()):
        assert abs(float(res.predicted_scales[h]) - float(o["skel_scales"][i])) < 2e-5
        assert np.abs(res.hand_poses[h].joint_angles - o["joint_angles"][i].numpy()).max() < 1e-4


def test_skin_landmarks_shim_leading_dims(labels, hand_model):
    from lib.common.hand_skinning import skin_landmarks
    ja = torch.from_numpy(labels["joint_angles"][:4].astype(np.float32))           # [4,2,22] on the CPU
    xf = torch.from_numpy(labels["wrist_transforms"][:4].astype(np.float32))
    out = skin_landmarks(hand_model, ja, xf)
    assert out.shape == (4, 2, 21, 3) and out.device.type == "cpu"
    want = ref_fk.skin_landmarks(_oracle_hm(labels), ja.numpy(), xf.numpy())
    assert np.abs(out.numpy() - want).max() < 2e-4
    one = skin_landmarks(hand_model, ja[2, 1].to(DEV), xf[2, 1].to(DEV))
    assert one.device.type == "cuda" and np.abs(one.cpu().numpy() - want[2, 1]).max() < 2e-4


@pytest.mark.parametrize("conv", ["fp32", "split_f16_always"])
@pytest.mark.parametrize("known", [True, False])
def test_batched_hot_path_vs_oracle(known, conv):
    """resample -> backbone -> head -> FK in one batched step against the oracle on the same crops, with the backbone's 3x3
    convolutions on the exact fp32 matrix instruction and on the split-fp16 kernels: the same tolerances (north_star)."""
    r = checks.run_small_end_to_end(synth.synthetic_state_dict(0), n_frames=3, device=DEV, known=known, conv=conv)
    assert r["hand_frames"] == 6
    assert r["crop_mismatch_fraction"] == 0.0 and r["crop_max_abs_diff"] == 0.0      # same cameras on both sides: the same crops
    assert r["max_joint_angle_err_rad"] < 1e-4
    assert r["max_wrist_translation_err_mm"] < 1e-3
    assert r["max_keypoint_err_mm"] < 1e-3


# ----------------------------------------------------------------------------- properties at full size
@pytest.fixture(scope="module")
def engine():
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    yield eng
    eng.close()


def test_full_size_batch_invariances(engine):
    """BASELINE config C2 size (256 frames x 1 hand x 2 views = 512 crops): results must not depend on how the
    batch is cut into passes or where a crop sits in the batch (bit-exact), and one sample must equal the
    same sample run alone."""
    n = 512
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    crops = torch.randint(0, 256, (n, 96, 96), device=DEV, generator=g).float() / 255.0
    base = engine.backbone(crops)
    assert torch.isfinite(base).all()
    for chunk in (64, 200):
        engine.set_backbone_chunk(chunk)
        assert torch.equal(engine.backbone(crops), base)
    engine.set_backbone_chunk(0)
    perm = torch.randperm(n, device=DEV, generator=g)
    assert torch.equal(engine.backbone(crops[perm]), base[perm])
    assert torch.equal(engine.backbone(crops[37:38]), base[37:38])
    # head: sample permutation with independent slots
    s = n // 2
    k = torch.eye(3, device=DEV).repeat(n, 1, 1)
    k[:, 0, 0] = k[:, 1, 1] = 100 + 60 * torch.rand(n, device=DEV, generator=g)
    k[:, 0, 2] = k[:, 1, 2] = 47.5
    x = torch.eye(4, device=DEV).repeat(n, 1, 1)
    x[:, :3, 3] = torch.rand(n, 3, device=DEV, generator=g) * 0.2
    sr = torch.arange(0, n, 2, device=DEV)[:, None] + torch.tensor([0, 2], device=DEV)
    hi = (torch.arange(s, device=DEV) % 2).long()
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    skel = torch.stack([hm.joint_rotation_axes.float(), hm.joint_rest_positions.float() * 0.001])[None].to(DEV)
    engine.reset_memory()
    pose, _ = engine.fuse_temporal_regress(base, k, x, sr, torch.arange(s, device=DEV), torch.zeros(s, dtype=torch.bool, device=DEV),
                                           hi, s, True, skel, _native.UT_MODE_KNOWN)
    pose = pose.clone()
    sp = torch.randperm(s, device=DEV, generator=g)
    crop_perm = (2 * sp[:, None] + torch.tensor([0, 1], device=DEV)).reshape(-1)
    engine.reset_memory()
    pose_p, _ = engine.fuse_temporal_regress(base[crop_perm], k[crop_perm], x[crop_perm], sr, torch.arange(s, device=DEV),
                                             torch.zeros(s, dtype=torch.bool, device=DEV), hi[sp], s, True, skel,
                                             _native.UT_MODE_KNOWN)
    assert torch.equal(pose_p, pose[sp])
    # decoded wrist transforms are rigid: R R^T = I, det = +-1 (x mirrored for right hands), last row 0 0 0 1
    xf = pose[:, 22:38].reshape(s, 4, 4).double()
    r = xf[:, :3, :3]
    assert (r @ r.transpose(1, 2) - torch.eye(3, device=DEV, dtype=torch.float64)).abs().max() < 1e-5
    det = torch.linalg.det(r)
    assert torch.allclose(det, torch.where(hi == 1, -1.0, 1.0).double(), atol=1e-5)
    assert torch.equal(xf[:, 3], torch.tensor([0, 0, 0, 1.0], device=DEV, dtype=torch.float64).expand(s, 4))
    assert (pose[:, 20:22] == 0).all() and (pose[:, 39:60] >= 1e-5).all()


def test_tracker_and_hotpath_share_one_handle(labels, hand_model):
    """A per-frame HandTracker (latency mode, deferred checks - for the duration of its own calls only) and a batched HotPath on
    the SAME native handle: the batch gives the records a fresh handle gives, bit for bit, before and after tracker calls,
    and the handle's modes are what they were."""
    from lib.models.umetrack_model import UmeTrackModel
    from lib.tracker.tracker import HandTracker, HandTrackerOpts
    sd = synth.synthetic_state_dict(0)
    model = UmeTrackModel(sd)
    model.eval()
    trk = HandTracker(model, HandTrackerOpts())
    eng = model.engine
    f = 24
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    src = torch.randint(0, 256, (f * 4, 480, 636), dtype=torch.uint8, device=DEV, generator=g)
    plan = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(labels, hand_model, range(f), DEV).items()}
    fresh = _native.HipEngine(sd, DEV)
    try:
        want = pipeline.HotPath(fresh, hand_model).step(pipeline.make_batch(plan, src, DEV)).clone()
    finally:
        fresh.close()
    hot = pipeline.HotPath(eng, hand_model)
    assert torch.equal(hot.step(pipeline.make_batch(plan, src, DEV)), want)
    assert (eng.deferred_checks, eng.latency_mode) == (False, False)
    frames = synth.synthetic_frames(1, seed=3)
    sample, cams = _input_frame(labels, 10, frames[0])
    cc = trk.gen_crop_cameras(cams, list(labels["camera_angles"]), hand_model, _gt(labels, 10), min_num_crops=1)
    res = trk.track_frame(sample, hand_model, cc)
    assert sorted(res.hand_poses) == [0, 1]
    assert (eng.deferred_checks, eng.latency_mode) == (False, False)
    model.reset_temporal_memory()
    assert torch.equal(hot.step(pipeline.make_batch(plan, src, DEV)), want)     # not the latency dispatch, no stale mode
    hot.check()


@pytest.mark.timeout(180, method="thread")
@pytest.mark.parametrize("conv,frames", [("fp32", 24), ("split_f16", 300)])
def test_whole_step_replays_from_one_hipgraph(labels, hand_model, conv, frames):
    """ut_warp_backbone + ut_fuse_temporal_regress + ut_fk (deferred index checks, workspace reserved by an eager step)
    captured into ONE hipGraph and replayed four times on alternating inputs: every replay equals the eager step on the same
    input and no check fires.  (Round 2's whole-path replay hung: the library zeroed its tile-queue words with hipMemsetAsync,
    and a captured memset node of >= 16 bytes fills with a stale pattern from the second replay on under ROCm 7.2
    (tools/diag/graph_memset.py); a negative queue word then walked a persistent kernel through ~10^9 tickets.  The words are
    zeroed by a kernel now, and the persistent loops compare tile indices as unsigned.)"""
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    try:
        eng.set_conv_arithmetic(conv)
        g = torch.Generator(device=DEV)
        g.manual_seed(3)
        src_a = torch.randint(0, 256, (frames * 4, 480, 636), dtype=torch.uint8, device=DEV, generator=g)
        src_b = torch.randint(0, 256, (frames * 4, 480, 636), dtype=torch.uint8, device=DEV, generator=g)
        plan = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(labels, hand_model, range(frames), DEV).items()}
        batch = pipeline.make_batch(plan, src_a.clone(), DEV)
        hot = pipeline.HotPath(eng, hand_model)
        want_a = hot.step(batch).clone()
        batch.src.copy_(src_b)
        want_b = hot.step(batch).clone()
        hot.check()
        assert not torch.equal(want_a, want_b)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(DEV)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                rec = hot.step(batch)
        for inp, want in ((src_a, want_a), (src_b, want_b), (src_a, want_a), (src_b, want_b)):
            batch.src.copy_(inp)
            rec.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(rec, want)
        hot.check()
        del graph
    finally:
        eng.close()


@pytest.mark.parametrize("conv", ["fp32", "split_f16"])
def test_c5_per_rank_workload_properties(conv):
    """BASELINE config C5 per rank (1024 frames x 4 cameras x 2 hands = 2048 hand-frames, 4096 crops - what one rank of
    the 8-GPU run processes per step, and bench.py's step): the fused path's records must be finite, must not depend
    on how the frames are cut into batches (each half run alone gives the same records, bit for bit: frames are
    independent, `memory_idx = arange`), the keypoints in a record must be the FK of that record's pose, wrist
    transforms rigid, and the unfused path (fp32 crops materialised) must give the same records.  In both arithmetics of
    the backbone (bench.py's default is split_f16): with the calibrated activation scales (the default) a crop's bits do not
    depend on its batch in split mode either, so an N-rank run reproduces the one-rank records bit for bit."""
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    eng.set_conv_arithmetic(conv)
    try:
        f = 1024
        g = torch.Generator(device=DEV)
        g.manual_seed(99)
        src = torch.randint(0, 256, (f * 4, 480, 636), dtype=torch.uint8, device=DEV, generator=g)
        plan = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(lab, hm, range(f), DEV).items()}
        hot = pipeline.HotPath(eng, hm)
        rec = hot.step(pipeline.make_batch(plan, src, DEV)).clone()
        hot.check()
        assert rec.shape == (2048, pipeline.RECORD) and torch.isfinite(rec).all()
        for lo, hi in ((0, 512), (512, 1024)):
            sub = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(lab, hm, range(lo, hi), DEV).items()}
            part = pipeline.HotPath(eng, hm).step(pipeline.make_batch(sub, src[lo * 4: hi * 4], DEV))
            assert torch.equal(part, rec[2 * lo: 2 * hi]), (lo, hi)
        unfused = pipeline.HotPath(eng, hm, keep_crops=True).step(pipeline.make_batch(plan, src, DEV))
        assert torch.equal(unfused, rec)
        blob = torch.from_numpy(_native.hand_model_blob(hm.joint_rotation_axes, hm.joint_rest_positions,
                                                        hm.landmark_rest_positions, hm.landmark_rest_bone_weights,
                                                        hm.landmark_rest_bone_indices)).reshape(1, 321).to(DEV)
        hand = torch.from_numpy(plan["hand_idx"]).to(DEV)
        kp = eng.fk(blob, rec[:, :22].contiguous(), rec[:, 22:38].reshape(-1, 4, 4).contiguous(), mirror=hand, t_scale=1000.0)
        assert torch.equal(kp.reshape(2048, -1), rec[:, 60:])
        r = rec[:, 22:38].reshape(-1, 4, 4)[:, :3, :3].double()
        assert (r @ r.transpose(1, 2) - torch.eye(3, device=DEV, dtype=torch.float64)).abs().max() < 1e-5
        # right hands: the x mirror of the crop camera and the un-mirror of the output cancel -> proper rotations
        assert torch.allclose(torch.linalg.det(r), torch.ones(2048, dtype=torch.float64, device=DEV), atol=1e-5)
    finally:
        eng.close()


def test_fk_rigid_equivariance_full_size(engine):
    """FK(T * wrist) == T * FK(wrist) for a rigid T, on 8192 x 2 poses (BASELINE config C5 record count)."""
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    blob = torch.from_numpy(_native.hand_model_blob(hm.joint_rotation_axes, hm.joint_rest_positions,
                                                    hm.landmark_rest_positions, hm.landmark_rest_bone_weights,
                                                    hm.landmark_rest_bone_indices)).reshape(1, 321).to(DEV)
    n = 16384
    idx = np.arange(n) % 369
    ja = torch.from_numpy(lab["joint_angles"][idx, 0].astype(np.float32)).to(DEV)
    xf = torch.from_numpy(lab["wrist_transforms"][idx, 0].astype(np.float32)).to(DEV)
    base = engine.fk(blob, ja, xf)
    ang = 0.7
    t = torch.tensor([[np.cos(ang), -np.sin(ang), 0, 12.0], [np.sin(ang), np.cos(ang), 0, -7.0], [0, 0, 1, 30.0], [0, 0, 0, 1]],
                     dtype=torch.float32, device=DEV)
    moved = engine.fk(blob, ja, t @ xf)
    want = base @ t[:3, :3].T + t[:3, 3]
    assert (moved - want).abs().max() < 2e-3          # mm; fp32 on ~500 mm coordinates
    assert torch.equal(engine.fk(blob, ja[:1], xf[:1]), base[:1])


def test_warp_identity_property(engine):
    """A crop camera that coincides with an (undistorted, pinhole-like) source camera resamples the centre of the
    source image exactly: with zero distortion and small angles the fisheye map is the arctan map, checked
    against the oracle on the full 480x636 source."""
    from absolutetrack_amd import geometry
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (1, 480, 636), dtype=np.uint8)
    cam = {"w": 636, "h": 480, "f": (240.0, 240.0), "c": (317.5, 239.5), "k": (0.0,) * 8, "T": np.eye(4)}
    crop = {"w": 96, "h": 96, "f": (240.0, 240.0), "c": (47.5, 47.5), "k": None, "T": np.eye(4)}
    want = ref_camera.warp_image(cam, crop, src[0], "cv2").astype(np.float32) / np.float32(255)
    got = engine.warp_crops(torch.from_numpy(src).to(DEV),
                            torch.from_numpy(geometry.pack_source_camera(cam["f"], cam["c"], cam["k"], cam["T"])[None]).to(DEV),
                            torch.from_numpy(geometry.pack_crop_camera(crop["f"], crop["c"], crop["T"])[None]).to(DEV),
                            torch.zeros(1, dtype=torch.int32, device=DEV)).cpu().numpy()[0]
    assert (got != want).mean() < 2e-3
    # the optical axis maps to the principal point: centre 2x2 block samples the 4 source pixels around (317.5,239.5)
    assert abs(got[47:49, 47:49].mean() * 255 - src[0, 239:241, 317:319].mean()) < 1.0


def test_batches_beyond_one_pass(engine):
    """More crops than one phase-A pass (4096) and than one phase-B pass (8192): the passes tile the batch and the
    result equals the per-slice results bit for bit; the chunk bound set by the 32-bit buffer offsets is enforced."""
    n = 8192 + 4096 + 37
    g = torch.Generator(device=DEV)
    g.manual_seed(11)
    crops = torch.randint(0, 256, (n, 96, 96), device=DEV, generator=g, dtype=torch.uint8).float() / 255.0
    full = engine.backbone(crops)
    assert torch.isfinite(full).all()
    for lo, hi in ((0, 64), (4090, 4100), (8185, 8200), (n - 37, n)):
        assert torch.equal(engine.backbone(crops[lo:hi]), full[lo:hi])
    with pytest.raises(RuntimeError):
        engine.set_backbone_chunk(7282)
    engine.set_backbone_chunk(7281)
    assert torch.equal(engine.backbone(crops[:7300]), full[:7300])
    engine.set_backbone_chunk(0)


def test_independent_processes_share_the_gpu():
    """The reference runs one model per Pool worker (run_eval_known_skeleton.py:117-119): independent processes,
    each with its own handle, must be able to use one GPU at the same time and get the single-process result."""
    import subprocess
    import sys
    code = ("import sys, torch; sys.path.insert(0, %r);"
            "from absolutetrack_amd import _native, pipeline, synth;"
            "lab = pipeline.load_labels(); hm = pipeline.hand_model_from_labels(lab);"
            "eng = _native.HipEngine(synth.synthetic_state_dict(0), 'cuda:0');"
            "plan = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(lab, hm, range(6), 'cuda:0').items()};"
            "src = torch.from_numpy(synth.synthetic_frames(6, seed=4).reshape(-1, 480, 636));"
            "hot = pipeline.HotPath(eng, hm);"
            "out = [hot.step(pipeline.make_batch(plan, src, 'cuda:0')).double().sum().item() for _ in range(20)];"
            "assert len(set(out)) == 1; print('SUM', repr(out[0]))") % ROOT_DIR
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(3)]
    sums = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        sums.append([ln for ln in so.splitlines() if ln.startswith("SUM")][0])
    assert len(set(sums)) == 1, sums


def test_batched_parity_legs_from_raw_images():
    """oracle/checks.py::batched_parity (the three legs every bench line reports) on 96 label frames, both arithmetics, with the
    bounds at what the runs measure: identical crop cameras -> identical crops (0 pixels) and every hand-frame inside 1e-4 rad /
    1e-3 mm; each side's OWN crop cameras with OpenCV's 8-bit remap -> at most 2e-4 of the pixels move by a grey level (the two
    sides' fp32 forward kinematics of the crop points differ in the last bit, and cv2 rounds coordinates to 1/32 px), the
    hand-frames whose crops are identical are inside tolerance; own cameras with the float remap (continuous in the coordinates)
    -> every hand-frame inside tolerance from raw images."""
    sd = synth.synthetic_state_dict(0)
    timed = checks.time_oracle(sd, 96, threads=min(16, os.cpu_count() or 1))
    for out in checks.batched_parity(sd, timed, DEV):
        n = out["hand_frames"]
        assert n == timed["hand_frames"] == 192
        a = out["identical_cameras"]
        assert a["crop_pixels_differing"] == 0 and a["hand_frames_outside_tolerance"] == 0, out
        assert a["max_joint_angle_err_rad"] < 1e-4 and a["max_keypoint_err_mm"] < 1e-3
        b = out["own_cameras_cv2"]
        assert b["crop_pixel_mismatch_fraction"] <= 2e-4, out
        assert b["hand_frames_with_identical_crops"] >= 1
        assert b["hand_frames_outside_tolerance"] <= n - b["hand_frames_with_identical_crops"], out
        c = out["own_cameras_float"]
        assert c["hand_frames_outside_tolerance"] == 0 and c["max_crop_abs_diff"] < 1e-4, out
