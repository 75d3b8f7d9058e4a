"""GPU parity of the batched crop-camera generator (ut_gen_crop_cameras) against
the reference-generated goldens (tests/golden/geometry_rec00.npz, made by oracle/gen_goldens.py from the reference's
lib.tracker.perspective_crop / lib.common.crop), against the CPU oracle on every label frame, and against the
per-frame host path it replaces.

Tolerances: the crop points come from fp32 FK (ours on the GPU, the reference's in torch fp32 on the CPU, equal to
~1e-4 mm), everything after it is fp64; a focal length therefore agrees to ~1e-6 relative and a rotation entry to
~1e-6.  Camera selection (integers) must be identical."""
import numpy as np
import pytest
import torch

from absolutetrack_amd import _native, arch, pipeline, synth
from oracle import ref_camera, scenarios

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CAM_FIELDS = ("ImageSizeX", "ImageSizeY", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "k5", "k6")


@pytest.fixture(scope="module")
def labels():
    return pipeline.load_labels()


@pytest.fixture(scope="module")
def hand_model(labels):
    return pipeline.hand_model_from_labels(labels)


def _run(labels, hand_model, frame_ids, wrist_xf=None, **kw):
    c = pipeline.label_candidates(labels, frame_ids)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    blob = torch.from_numpy(_native.hand_model_blob(
        hand_model.joint_rotation_axes, hand_model.joint_rest_positions, hand_model.landmark_rest_positions,
        hand_model.landmark_rest_bone_weights, hand_model.landmark_rest_bone_indices)).reshape(1, 321).to(DEV)
    args = dict(max_views=2, min_vis=19, crop_size=arch.CROP, focal_multiplier=0.8)
    args.update(kw)
    g = _native.gen_crop_cameras(t(c["cam_params"]), t(c["camera_angles"]), blob, hand_model.joint_limits.float().to(DEV),
                                 t(c["joint_angles"]), t(c["wrist_xf"] if wrist_xf is None else wrist_xf), t(c["frame_idx"]),
                                 t(c["hand_idx"]), c["n_cams"], c["src_wh"], **args)
    torch.cuda.synchronize()
    return c, {k: v.cpu().numpy() for k, v in g.items()}


def _crop_T(row):
    t = np.eye(4)
    t[:3, :3] = row[4:13].reshape(3, 3)
    t[:3, 3] = row[13:16]
    return t


def test_matches_reference_goldens(labels, hand_model, golden_dir):
    g = dict(np.load(f"{golden_dir}/geometry_rec00.npz"))
    frames = [int(f) for f in g["frames"]]
    c, out = _run(labels, hand_model, frames)
    assert out["status"].max() == 0
    n = 0
    for i, (fi, hand) in enumerate(zip(c["frame_idx"], c["hand_idx"])):
        key = f"f{frames[fi]}.h{hand}."
        cams = list(g[key + "cams"])
        assert out["n_views"][i] == len(cams)
        assert list(out["cam_index"][i, : len(cams)]) == cams
        for v, ci in enumerate(cams):
            ck = key + f"c{ci}."
            row = out["crop_params"][i, v]
            np.testing.assert_allclose(row[0:2], g[ck + "f"], rtol=5e-6)
            np.testing.assert_allclose(row[2:4], g[ck + "c"], rtol=0)
            T = _crop_T(row)
            np.testing.assert_allclose(T[:3, :3], g[ck + "T"][:3, :3], atol=5e-6)
            np.testing.assert_allclose(T[:3, 3], g[ck + "T"][:3, 3], atol=1e-6)      # mm; the camera does not move
            np.testing.assert_allclose(out["intrinsics"][i, v], g[ck + "K"], rtol=5e-6)
            n += 1
    assert n == 40


def test_matches_oracle_on_every_label_frame(labels, hand_model):
    """All 369 label frames x 2 hands in one launch vs the CPU oracle (itself pinned to the goldens)."""
    n_lab = labels["joint_angles"].shape[0]
    c, out = _run(labels, hand_model, range(n_lab))
    hm = scenarios.hand_model_mm()
    worst_f = worst_r = worst_e = 0.0
    for i, (fi, hand) in enumerate(zip(c["frame_idx"], c["hand_idx"])):
        cams = [ref_camera.camera_from_json(dict(zip(CAM_FIELDS, labels["cameras"][ci])) | {"DistortionModel": "FishEye62"},
                                            labels["camera_to_world_transforms"][fi, ci]) for ci in range(4)]
        crops = ref_camera.gen_crop_cameras(cams, labels["camera_angles"], hm, labels["joint_angles"][fi, hand],
                                            labels["wrist_transforms"][fi, hand], int(hand))
        assert list(out["cam_index"][i, : out["n_views"][i]]) == list(crops), (fi, hand)
        assert (out["cam_index"][i, out["n_views"][i]:] == -1).all()
        for v, (ci, cc) in enumerate(crops.items()):
            row = out["crop_params"][i, v]
            worst_f = max(worst_f, abs(row[0] / cc["f"][0] - 1))
            worst_r = max(worst_r, np.abs(_crop_T(row) - cc["T"])[:3, :3].max())
            k, ext = ref_camera.network_inputs_for_crop(cc)
            worst_e = max(worst_e, np.abs(out["extrinsics"][i, v] - ext).max())
            assert row[1] == row[0] and row[2] == row[3] == (arch.CROP - 1) / 2
    assert out["status"].max() == 0
    assert worst_f < 5e-6 and worst_r < 5e-6 and worst_e < 5e-6, (worst_f, worst_r, worst_e)


def test_device_plan_equals_host_plan_and_feeds_the_hot_path(labels, hand_model):
    frames = list(range(0, 369, 7))
    host = pipeline.crop_plan_from_labels(labels, hand_model, frames)
    dev = pipeline.crop_plan_on_device(labels, hand_model, frames, DEV)
    for k in ("src_index", "sample_range", "hand_idx"):
        assert np.array_equal(dev[k].cpu().numpy(), host[k]), k
    np.testing.assert_allclose(dev["cam_params"].cpu().numpy(), host["cam_params"], rtol=0, atol=0)
    np.testing.assert_allclose(dev["crop_params"].cpu().numpy(), host["crop_params"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(dev["intrinsics"].cpu().numpy(), host["intrinsics"], rtol=2e-6)
    np.testing.assert_allclose(dev["extrinsics"].cpu().numpy(), host["extrinsics"], rtol=0, atol=2e-6)
    # the two plans drive the hot path to the same poses
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    hp = pipeline.HotPath(eng, hand_model)
    src = torch.from_numpy(synth.synthetic_frames(len(frames), seed=3).reshape(-1, 480, 636))
    rec_h = hp.step(pipeline.make_batch(host, src, DEV)).clone()
    dplan = {k: v.cpu().numpy() for k, v in dev.items()}
    rec_d = hp.step(pipeline.make_batch(dplan, src, DEV)).clone()
    torch.cuda.synchronize()
    d = (rec_h - rec_d).abs()
    assert float(d[:, :22].max()) < 2e-3          # rad: crops resampled from cameras equal to ~1e-6
    assert float(d[:, arch.POSE_REC:].max()) < 0.5  # mm


def test_no_eligible_view_and_unbuildable_crop(labels, hand_model):
    # a hand one metre behind the headset (opposite the cameras' mean optical axis): no eligible view, slots stay -1
    from absolutetrack_amd.tracker import SingleHandPose, _visible_counts, gen_crop_cameras_from_pose, landmarks_from_hand_pose
    c = pipeline.label_candidates(labels, [0])
    c2w = labels["camera_to_world_transforms"][0]
    behind = c2w[:, :3, 3].mean(0) - 1000.0 * c2w[:, :3, 2].mean(0) / np.linalg.norm(c2w[:, :3, 2].mean(0))
    far = c["wrist_xf"].copy()
    far[:, :3, 3] = behind.astype(np.float32)
    cams = pipeline.cameras_for_frame(labels, 0)
    for h in (0, 1):      # precondition, from the host path
        pose = SingleHandPose(joint_angles=labels["joint_angles"][0, h], wrist_xform=far[h], hand_confidence=1.0)
        assert max(_visible_counts(cams, landmarks_from_hand_pose(hand_model, pose, h))) < 19
    _, out = _run(labels, hand_model, [0], wrist_xf=far)
    assert (out["n_views"] == 0).all() and (out["cam_index"] == -1).all() and (out["status"] == 0).all()
    # a wrist sitting in camera 0's centre with the visibility gate off: crop points fall behind the crop camera;
    # the host path raises ValueError("Unable to create crop camera"), the kernel flags status 1
    at_cam = c["wrist_xf"].copy()
    at_cam[:, :3, 3] = labels["camera_to_world_transforms"][0, 0, :3, 3].astype(np.float32)
    _, out = _run(labels, hand_model, [0], wrist_xf=at_cam, min_vis=0)
    assert (out["status"] == 1).all()
    pose = SingleHandPose(joint_angles=labels["joint_angles"][0, 0], wrist_xform=at_cam[0], hand_confidence=1.0)
    with pytest.raises(ValueError):
        gen_crop_cameras_from_pose(cams, labels["camera_angles"], hand_model, pose, 0, 63, np.array([96, 96]),
                                   max_view_num=2, sort_camera_index=True, focal_multiplier=0.8,
                                   min_required_vis_landmarks=0)


def test_landmarks_output_equals_fk(labels, hand_model):
    """The optional landmarks output (the label pose's first 21 crop points) is ut_fk of the same pose, bit for bit -
    what the per-frame tracker hands to landmarks_from_hand_pose instead of a second launch."""
    c = pipeline.label_candidates(labels, list(range(0, 369, 7)))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    blob = torch.from_numpy(_native.hand_model_blob(
        hand_model.joint_rotation_axes, hand_model.joint_rest_positions, hand_model.landmark_rest_positions,
        hand_model.landmark_rest_bone_weights, hand_model.landmark_rest_bone_indices)).reshape(1, 321).to(DEV)
    g = _native.gen_crop_cameras(t(c["cam_params"]), t(c["camera_angles"]), blob, hand_model.joint_limits.float().to(DEV),
                                 t(c["joint_angles"]), t(c["wrist_xf"]), t(c["frame_idx"]), t(c["hand_idx"]), c["n_cams"],
                                 c["src_wh"], want_landmarks=True)
    want = _native.fk_stateless(blob, t(c["joint_angles"]), t(c["wrist_xf"]), mirror=t(c["hand_idx"]))
    assert g["landmarks"].shape == want.shape and torch.equal(g["landmarks"], want)


def test_argument_checks(labels, hand_model):
    c = pipeline.label_candidates(labels, [0])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    blob = torch.zeros(1, 321, device=DEV)
    lim = hand_model.joint_limits.float().to(DEV)
    with pytest.raises(ValueError):      # frame index past the camera rows
        _native.gen_crop_cameras(t(c["cam_params"]), t(c["camera_angles"]), blob, lim, t(c["joint_angles"]),
                                 t(c["wrist_xf"]), t(c["frame_idx"] + 5), t(c["hand_idx"]), 4, c["src_wh"])
    with pytest.raises(ValueError):      # n disagreement
        _native.gen_crop_cameras(t(c["cam_params"]), t(c["camera_angles"]), blob, lim, t(c["joint_angles"][:1]),
                                 t(c["wrist_xf"]), t(c["frame_idx"]), t(c["hand_idx"]), 4, c["src_wh"])
    with pytest.raises(_native.NativeLibraryError):   # no CPU fallback
        cpu = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        _native.gen_crop_cameras(cpu(c["cam_params"]), cpu(c["camera_angles"]), blob.cpu(), lim.cpu(),
                                 cpu(c["joint_angles"]), cpu(c["wrist_xf"]), cpu(c["frame_idx"]), cpu(c["hand_idx"]), 4,
                                 c["src_wh"])
    lib = _native.load_library()
    assert lib.ut_gen_crop_cameras(None, None, None, None, None, 1, None, None, None, None, 4, 4, 2, 19, 636, 480, 96,
                                   0.8, None, None, None, None, None, None, None, None) != 0
    assert b"ut_gen_crop_cameras" in lib.ut_last_error(None)


def test_planner_refreshes_a_batch_in_place(labels, hand_model):
    """DeviceCropPlanner (crop cameras regenerated inside the step, no host round trip) rewrites exactly what
    crop_plan_on_device produced for the same frames."""
    frames = list(range(20, 60))
    plan = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(labels, hand_model, frames, DEV).items()}
    src = torch.zeros(len(frames) * 4, 480, 636, dtype=torch.uint8)
    batch = pipeline.make_batch(plan, src, DEV)
    want = {k: getattr(batch, k).clone() for k in ("crop_params", "intrinsics", "extrinsics", "src_index")}
    for k in want:
        getattr(batch, k).zero_()
    planner = pipeline.DeviceCropPlanner(labels, hand_model, frames, DEV)
    planner.refresh(batch)
    torch.cuda.synchronize()
    assert bool(planner.ok.item())
    for k, w in want.items():
        assert torch.equal(getattr(batch, k), w), k
    # a pose no camera sees: the flag drops, nothing faults
    c2w = labels["camera_to_world_transforms"][frames[0]]
    fwd = c2w[:, :3, 2].mean(0)
    behind = c2w[:, :3, 3].mean(0) - 1000.0 * fwd / np.linalg.norm(fwd)
    planner.wrist_xf[0, :3, 3] = torch.from_numpy(behind.astype(np.float32)).to(DEV)
    planner.refresh(batch)
    assert not bool(planner.ok.item())
