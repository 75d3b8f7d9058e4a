"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host-side mirror
of the reference interface (geometry, hand model helpers, bundles, fs, model container, sharding) behaves
like the reference, and the product refuses to run its hot path without a HIP device."""
import os
import re

import numpy as np
import pytest
import torch

from absolutetrack_amd import _native, arch, bundles, geometry, hand, model, pipeline, synth, tracker
from oracle import ref_camera, scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = not torch.cuda.is_available()


# ----------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "umetrack_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ut_[a-z_]+)\s*\(", header)))
    assert declared == sorted(_native.EXPORTS)
    lib = _native.load_library()
    assert os.path.samefile(lib._name, _native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ut_weight_blob_floats() == 4_259_410       # no compute, no device needed


def test_state_dict_blob_is_strict():
    sd = synth.synthetic_state_dict(0)
    blob = _native.state_dict_to_blob(sd)
    assert blob.dtype == np.float32 and blob.size == 4_259_410
    first = arch.state_dict_spec()[0][0]
    assert np.array_equal(blob[: 32 * 9], sd[first].reshape(-1))
    bad = dict(sd)
    bad.pop(first)
    with pytest.raises(RuntimeError):
        _native.state_dict_to_blob(bad)
    bad = dict(sd)
    bad[first] = np.zeros((32, 1, 5, 5), np.float32)
    with pytest.raises(RuntimeError):
        _native.state_dict_to_blob(bad)


def test_load_pretrained_model_round_trip(tmp_path):
    """lib/models/model_loader.py:84-87: a plain `torch.save`d state dict -> file -> UmeTrackModel, strict.  The file
    is read with weights_only=True; a checkpoint with a missing / extra / mis-shaped entry is refused like
    load_state_dict(strict=True) refuses it."""
    from lib.models.model_loader import load_pretrained_model
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synthetic_state_dict(0).items()}
    path = str(tmp_path / "pretrained_weights.torch")
    torch.save(sd, path)
    m = load_pretrained_model(path)
    assert isinstance(m, model.UmeTrackModel) and m.eval() is m
    got = m.state_dict()
    assert list(got) == [k for k, _s, _kind in arch.state_dict_spec()]
    assert all(torch.equal(got[k], sd[k]) and got[k].dtype == sd[k].dtype for k in sd)
    assert np.array_equal(_native.state_dict_to_blob(got), _native.state_dict_to_blob(synth.synthetic_state_dict(0)))
    first = arch.state_dict_spec()[0][0]
    for bad in ({k: v for k, v in sd.items() if k != first}, dict(sd, extra=torch.zeros(1)),
                dict(sd, **{first: torch.zeros(32, 1, 5, 5)})):
        torch.save(bad, path)
        with pytest.raises(RuntimeError):
            load_pretrained_model(path)
    # an object that needs unpickling of arbitrary code is refused by the safe loader
    import pickle
    with open(path, "wb") as f:
        pickle.dump({"w": np.zeros(3)}, f)
    with pytest.raises(Exception):
        load_pretrained_model(path)


@pytest.mark.skipif(not NO_GPU, reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    m = model.UmeTrackModel(synth.synthetic_state_dict(0))
    assert m.getInputImageSizes() == (96, 96)
    fd = model.InputFrameData(torch.zeros(2, 96, 96), torch.eye(3).repeat(2, 1, 1), torch.eye(4).repeat(2, 1, 1))
    desc = model.InputFrameDesc(torch.tensor([[0, 2]]), torch.tensor([0]), torch.tensor([False]), torch.tensor([0]))
    with pytest.raises(_native.NativeLibraryError):
        m.regress_pose_pred_skel_scale(fd, desc)
    with pytest.raises(_native.NativeLibraryError):
        _native.HipEngine(synth.synthetic_state_dict(0), "cuda")
    hm = pipeline.hand_model_from_labels(pipeline.load_labels())
    with pytest.raises(_native.NativeLibraryError):
        hand.skin_landmarks(hm, torch.zeros(22), torch.eye(4))


def test_model_container_surface():
    sd = {k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(0).items()}
    m = model.UmeTrackModel()
    m.load_state_dict(sd)
    assert list(m.state_dict()) == [k for k, _s, _k in arch.state_dict_spec()]
    assert m.eval() is m and m.to("cpu") is m
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in list(sd.items())[:-1]})
    with pytest.raises(RuntimeError):
        m.train()


# ----------------------------------------------------------------------------- geometry vs reference goldens
def _cams(lab, fi):
    return pipeline.cameras_for_frame(lab, fi)


def test_product_geometry_matches_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, "geometry_rec00.npz"))
    lab = pipeline.load_labels()
    for fi in g["frames"]:
        cams = _cams(lab, int(fi))
        for hnd in (0, 1):
            key = f"f{fi}.h{hnd}."
            pts = g[key + "crop_points"]
            counts = tracker._visible_counts(cams, pts[:21])
            assert counts == list(g[key + "visible"])
            for ci in g[key + "cams"]:
                cc = geometry.gen_crop_parameters_from_points(cams[ci], pts, (96, 96), mirror_img_x=(hnd == 1),
                                                              camera_angle=lab["camera_angles"][ci], focal_multiplier=0.8)
                ck = key + f"c{ci}."
                np.testing.assert_allclose(cc.f, g[ck + "f"], rtol=1e-12)
                np.testing.assert_allclose(cc.c, g[ck + "c"], rtol=0)
                np.testing.assert_allclose(cc.camera_to_world_xf, g[ck + "T"], atol=1e-10)
                np.testing.assert_allclose(cc.uv_to_window_matrix(), g[ck + "K"], rtol=1e-12)
                k, ext = tracker.network_camera_inputs(cc)
                ko, eo = ref_camera.network_inputs_for_crop({"f": cc.f, "c": cc.c, "T": cc.camera_to_world_xf})
                np.testing.assert_allclose(k, ko, rtol=1e-6)
                np.testing.assert_allclose(ext, eo, rtol=1e-6, atol=1e-9)
                # host camera maths (the same the warp kernel evaluates per pixel) vs the reference's map
                px, py = np.meshgrid(np.arange(0, 96, 4), np.arange(0, 96, 4))
                dst = np.column_stack((px.ravel(), py.ravel()))
                eye = cams[ci].world_to_eye(cc.eye_to_world(cc.window_to_eye(dst)))
                win = cams[ci].eye_to_window(eye)
                win[eye[:, 2] < 0] = -1
                np.testing.assert_allclose(win.astype(np.float32).reshape(24, 24, 2), g[ck + "map_sub"], atol=2e-4)
                row = geometry.pack_camera_model(cc)
                assert row.shape == (24,) and row[0] == cc.f[0] and row[15] == cc.camera_to_world_xf[2, 3]
                srow = geometry.pack_camera_model(cams[ci])
                assert srow.shape == (32,) and np.array_equal(srow[4:12], np.array(tuple(cams[ci].distort)))


def test_crop_camera_errors_like_reference():
    lab = pipeline.load_labels()
    cam = _cams(lab, 0)[0]
    behind = np.array([[0.0, 0.0, -100.0], [10.0, 0.0, -120.0]]) @ cam.camera_to_world_xf[:3, :3].T + cam.camera_to_world_xf[:3, 3]
    with pytest.raises(ValueError):
        geometry.gen_intrinsics_from_bounding_pts(np.array([[0.0, 0.0, -1.0]]), 96, 96)
    with pytest.raises(ValueError):       # a point set that needs a focal < 5 px
        geometry.gen_intrinsics_from_bounding_pts(np.array([[100.0, 0.0, 1.0]]), 96, 96)
    del behind


def test_camera_json_and_copy():
    js = {"ImageSizeX": 636, "ImageSizeY": 480, "fx": 235.9, "fy": 235.8, "cx": 317.3, "cy": 240.1,
          "DistortionModel": "FishEye62", "k1": -0.02, "k2": 0.1, "k3": -0.07, "k4": 0.01, "p1": -2e-4, "p2": -1e-3,
          "k5": 3e-3, "k6": -7e-4}
    cam = geometry.read_camera_from_json(js)
    assert isinstance(cam, geometry.Fisheye62CameraModel) and tuple(cam.distort)[4] == -2e-4
    t = np.eye(4)
    t[:3, 3] = (1, 2, 3)
    c2 = cam.copy(camera_to_world_xf=t)
    assert c2.width == 636 and np.allclose(c2.c, cam.c) and c2.camera_to_world_xf is t
    v = np.array([[0.1, -0.2, 1.0], [0.3, 0.1, 0.8]])
    np.testing.assert_allclose(c2.world_to_eye(c2.eye_to_world(v)), v, atol=1e-12)
    oc = {"w": 636, "h": 480, "f": cam.f, "c": cam.c, "k": tuple(cam.distort), "T": t}
    np.testing.assert_allclose(cam.eye_to_window(v), ref_camera.eye_to_window(oc, v), rtol=1e-12)
    pin = geometry.read_camera_from_json({**js, "DistortionModel": "PinholePlane"})
    np.testing.assert_allclose(pin.eye_to_window(pin.window_to_eye(np.array([[10.0, 20.0]]))), [[10.0, 20.0]], atol=1e-9)


# ----------------------------------------------------------------------------- hand model helpers
def test_scaled_and_mirrored_hand_model():
    hm = pipeline.hand_model_from_labels(pipeline.load_labels())
    s = hand.scaled_hand_model(hm, 0.001)
    assert torch.allclose(s.joint_rest_positions, hm.joint_rest_positions * 0.001)
    assert torch.allclose(s.landmark_rest_positions, hm.landmark_rest_positions * 0.001)
    assert s.joint_rotation_axes is hm.joint_rotation_axes
    batched = hm._replace(joint_rotation_axes=hm.joint_rotation_axes[None].repeat(3, 1, 1),
                          joint_rest_positions=hm.joint_rest_positions[None].repeat(3, 1, 1),
                          landmark_rest_positions=hm.landmark_rest_positions[None].repeat(3, 1, 1))
    m = hand.mirrored_hand_model(batched, torch.tensor([False, True, False]))
    assert torch.equal(m.joint_rest_positions[0], hm.joint_rest_positions)
    assert torch.equal(m.joint_rest_positions[1, :, 0], -hm.joint_rest_positions[:, 0])
    assert torch.equal(m.joint_rest_positions[1, :, 1:], hm.joint_rest_positions[:, 1:])
    assert torch.equal(m.joint_rotation_axes[1, :, 0], hm.joint_rotation_axes[:, 0])
    assert torch.equal(m.joint_rotation_axes[1, :, 1:], -hm.joint_rotation_axes[:, 1:])
    assert torch.equal(m.landmark_rest_positions[1, :, 0], -hm.landmark_rest_positions[:, 0])
    assert hand.NUM_JOINT_FRAMES == 17 and hand.NUM_LANDMARKS_PER_HAND == 21
    blob = _native.hand_model_blob(hm.joint_rotation_axes, hm.joint_rest_positions, hm.landmark_rest_positions,
                                   hm.landmark_rest_bone_weights, hm.landmark_rest_bone_indices)
    assert blob.shape == (321,) and blob[258] == float(hm.landmark_rest_bone_indices[0, 0])


# ----------------------------------------------------------------------------- bundles / fs
def test_bundles_helpers():
    out = [model.RegressorOutput(torch.full((2, 22), float(i)), torch.zeros(2, 4, 4), None, torch.ones(2, 21)) for i in range(3)]
    c = bundles.collate(out)
    assert isinstance(c, model.RegressorOutput) and c.joint_angles.shape == (3, 2, 22) and c.skel_scales is None
    t = bundles.map_fields(lambda x: x.transpose(0, 1) if x is not None else None, c)
    assert t.joint_angles.shape == (2, 3, 22)
    moved = bundles.to_device((c, {"a": torch.zeros(1)}, [np.zeros(2)]), torch.device("cpu"))
    assert moved[1]["a"].device.type == "cpu" and isinstance(moved[2][0], np.ndarray)
    with pytest.raises(TypeError):
        bundles.collate([None, torch.zeros(1)])
    import lib.data_utils.fs as fs
    assert fs.join("/a/b", "/c/d.npy") == "/a/b/c/d.npy" and fs.join("a", "") == "a" and fs.dirname("/a/b/c.mp4") == "/a/b"


def test_lib_shims_expose_reference_surface():
    import lib.common.camera as cam
    import lib.common.crop as crop
    import lib.common.hand as lhand
    import lib.common.hand_skinning as skin
    import lib.models.model_loader as loader
    import lib.models.regressor as reg
    import lib.models.umetrack_model as um
    import lib.tracker.perspective_crop as pc
    import lib.tracker.tracker as trk
    import lib.tracker.tracking_result as tr
    assert um.UmeTrackModel is model.UmeTrackModel and reg.RegressorOutput is model.RegressorOutput
    assert trk.HandTracker is tracker.HandTracker and trk.MAX_VIEW_NUM == 2 and trk.HandTrackerOpts().hand_ratio_in_crop == 0.8
    assert pc.landmarks_from_hand_pose is tracker.landmarks_from_hand_pose and tr.SingleHandPose is tracker.SingleHandPose
    assert cam.Fisheye62CameraModel is geometry.Fisheye62CameraModel and crop.gen_crop_parameters_from_points
    assert lhand.NUM_HANDS == 2 and skin.skin_landmarks is hand.skin_landmarks and callable(loader.load_pretrained_model)
    assert tr.TrackingResult().hand_poses == {}


# ----------------------------------------------------------------------------- sharding
def test_shard_frames_partitions_exactly():
    for total, world in ((8192, 8), (10, 4), (7, 2), (3, 8)):
        spans = [pipeline.shard_frames(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_shard_sequences_keeps_slots_rank_local():
    """Sequence mode: a rank owns the same contiguous sequence block at every time step, a hand-sample keeps one local
    slot, memory is used from the second step on (SURVEY.md section 8 e; lib/models/temporal.py:101-137)."""
    import torch
    for total, world in ((2048, 8), (10, 3), (5, 2)):
        spans = [pipeline.shard_sequences(total, r, world) for r in range(world)]
        assert spans == [pipeline.shard_frames(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
    hand = torch.tensor([0, 1, 0, 1, 0, 1])
    m0, u0, n0 = pipeline.sequence_step_descriptors(hand, first_step=True)
    m1, u1, n1 = pipeline.sequence_step_descriptors(hand, first_step=False)
    assert m0.tolist() == list(range(6)) and torch.equal(m0, m1) and n0 == n1 == 6
    assert u0.tolist() == [0] * 6 and u1.tolist() == [1] * 6 and u0.dtype == torch.uint8


def test_synthetic_inputs_are_reproducible():
    a, b = synth.synthetic_crops(3, seed=4), synth.synthetic_crops(3, seed=4)
    assert np.array_equal(a, b) and a.dtype == np.float32 and 0 <= a.min() and a.max() <= 1
    assert np.all(np.abs(a * 255 - np.rint(a * 255)) < 1e-4)          # exactly representable u8/255 values
    f = synth.synthetic_frames(1, seed=2)
    assert f.shape == (1, 4, 480, 636) and f.dtype == np.uint8 and f.std() > 20
    st = scenarios.model_steps(True)
    assert [s["sample_range"].tolist() for s in st][0] == [[0, 2], [2, 3], [3, 5]]


def test_torch_data_time_step_batching_matches_oracle():
    """run_inference_torch_data.py:39-85 - host index bookkeeping, no GPU involved."""
    import torch
    from absolutetrack_amd import torch_data as td
    from absolutetrack_amd.hand import HandModel
    from oracle import ref_torch_data as rt
    rng = np.random.default_rng(3)
    bs, seq = 3, 4
    img = rng.random((bs, seq, 2, 8, 8), dtype=np.float32)
    k = rng.random((bs, seq, 2, 3, 3), dtype=np.float32)
    x = rng.random((bs, seq, 2, 4, 4), dtype=np.float32)
    hand = np.repeat(np.array([[0.0], [1.0], [1.0]], np.float32), seq, 1)
    axes = rng.random((bs, seq, 22, 3), dtype=np.float32)
    rest = rng.random((bs, seq, 22, 3), dtype=np.float32)
    z = torch.zeros(1)
    hm = HandModel(joint_rotation_axes=torch.from_numpy(axes), joint_rest_positions=torch.from_numpy(rest),
                   joint_frame_index=z, joint_parent=z, joint_first_child=z, joint_next_sibling=z,
                   landmark_rest_positions=z, landmark_rest_bone_weights=z, landmark_rest_bone_indices=z, hand_scale=None)
    pose = td.PoseData(joint_angles=z, wrist_xfs=z, left_hand_model=hm)
    mi = td.ModelInput(orig_pose_data=pose, s_solved_pose_data=pose, left_images=torch.from_numpy(img),
                       intrinsics=torch.from_numpy(k), extrinsics_xf=torch.from_numpy(x), hand_idx=torch.from_numpy(hand))
    for mode in ("multiv", "singlev"):
        want = rt.unpack_batched_data(img, k, x, hand, axes, rest, mode)
        got = td.unpack_batched_data(mi, mode)
        assert len(got) == len(want) == seq
        for (fd, desc, sk), w in zip(got, want):
            assert np.array_equal(fd.left_images.numpy(), w["images"])
            assert np.array_equal(fd.intrinsics.numpy(), w["intrinsics"])
            assert np.array_equal(fd.extrinsics_xf.numpy(), w["extrinsics"])
            assert np.array_equal(desc.sample_range.numpy(), w["sample_range"]) and desc.sample_range.dtype == torch.int64
            assert np.array_equal(desc.memory_idx.numpy(), w["memory_idx"])
            assert np.array_equal(desc.use_memory.numpy(), w["use_memory"]) and desc.use_memory.dtype == torch.bool
            assert np.array_equal(desc.hand_idx.numpy(), w["hand_idx"]) and desc.hand_idx.dtype == torch.int64
            assert np.array_equal(sk.joint_rotation_axes.numpy(), w["axes"])
            assert np.array_equal(sk.joint_rest_positions.numpy(), w["rest"])
    with pytest.raises(ValueError):
        td.unpack_batched_data(mi, "stereo")
    with pytest.raises(ValueError):
        rt.unpack_batched_data(img, k, x, hand, axes, rest, "stereo")


def test_boundary_header_is_plain_c_and_callable_from_c(tmp_path):
    """include/umetrack_hip.h compiles as C99 and the library's entry points work through C function pointers
    (argument validation only - no GPU call)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists(_native.LIB_PATH):
        pytest.skip("needs gcc and the built library")
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_check.c"), "-o", exe, "-ldl"])
    out = subprocess.run([exe, _native.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert int(out.stdout.strip()) == sum(int(np.prod(s)) for _k, s, _kind in arch.state_dict_spec())


def test_landmark_memo_is_exact_and_invalidates():
    """The per-frame tracker remembers FK results for landmarks_from_hand_pose (tracker._LandmarkMemo): an entry is
    served only for the same model tensors (same objects, same in-place version), the same hand and bit-identical pose
    arrays (compared as float32, the precision the FK kernel sees)."""
    from absolutetrack_amd import tracker as tk
    hm = pipeline.hand_model_from_labels(pipeline.load_labels())
    m = tk._LandmarkMemo(cap=2)
    ja, xf = np.linspace(0, 1, 22), np.eye(4)
    kp = np.arange(63, dtype=np.float32).reshape(21, 3)
    m.put(hm, 1, ja, xf, kp)
    got = m.get(hm, 1, ja.astype(np.float32), xf.astype(np.float32))
    assert np.array_equal(got, kp) and got is not kp
    assert m.get(hm, 0, ja, xf) is None                                   # other hand
    assert m.get(hm, 1, ja + 1e-6, xf) is None                            # other pose
    other = hm._replace(joint_rest_positions=hm.joint_rest_positions.clone())
    assert m.get(other, 1, ja, xf) is None                                # other model tensors
    hm.joint_rest_positions.mul_(1.0)                                     # same object, new in-place version
    assert m.get(hm, 1, ja, xf) is None
    m.put(hm, 1, ja, xf, kp); m.put(hm, 0, ja, xf, kp); m.put(hm, 1, ja + 1, xf, kp)
    assert len(m.items) == 2 and m.get(hm, 1, ja, xf) is None             # capacity: oldest entry dropped


@pytest.mark.parametrize("source", ["conv_c64k.hip", "conv_c32s2.hip"])
def test_register_resident_kernels_keep_their_register_plan(tmp_path, source):
    """conv_c64k.hip / conv_c32s2.hip keep 32 weight fragments pinned in the accumulator half of the register file and are written to
    need no scratch: a compiler that moves a pinned fragment (v_accvgpr_*), spills (scratch_*) or gives the kernel fewer than two
    waves per SIMD has broken the plan the kernels' speed - and, for values parked in registers the weights live in, their results -
    rest on.  Also: no timing / ablation code in the product sources (tools/diag patches a copy)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.abspath(_native.__file__)), "csrc")
    text = open(os.path.join(csrc, source)).read()
    assert "s_memtime" not in text and "STAMP" not in text and "ABL" not in text
    out = tmp_path / "k.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-I", csrc, os.path.join(csrc, source), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert "v_accvgpr" not in asm
    assert "scratch_" not in asm
    assert asm.count("v_mfma_f32_32x32x16_f16") in (108, 60)          # per wave and tile: 9 x 2 x 2 x 3, or 9 x 2 x 3 + 2 x 3
    assert re.search(r"Occupancy \[waves/SIMD\]: 2", r.stderr)


def _parse_canonical(flat):
    """ut_canonical_backbone_weights' layout -> (stem, [(conv1, conv2, ds | None, stride)], proj), each conv = (w OIHW, b)."""
    pos = 0

    def take(cout, cin, k):
        nonlocal pos
        w = flat[pos: pos + cout * cin * k * k].reshape(cout, cin, k, k); pos += w.size
        b = flat[pos: pos + cout]; pos += cout
        return torch.from_numpy(w.copy()), torch.from_numpy(b.copy())
    stem = take(32, 1, 3)
    blocks = []
    for _p, cin, cout, stride, ds in arch.backbone_blocks():
        c1, c2 = take(cout, cin, 3), take(cout, cout, 3)
        blocks.append((c1, c2, take(cout, cin, 1) if ds else None, stride))
    proj = take(72, 256, 1)
    assert pos == flat.size
    return stem, blocks, proj


@pytest.mark.parametrize("spread", [8, 24])
def test_pack_time_channel_canonicalisation_is_exact_and_invariant(spread):
    """ut_create brings every inner and trunk channel of the backbone to a canonical power-of-two scale before packing
    (csrc/ut_api.hip::fold_backbone; host only, so it is checked here without a GPU through ut_canonical_backbone_weights):
    (i) checkpoints that differ by per-channel powers of two - up to 2^48 between two channels of one tensor here - pack to the
    same tensors bit for bit, which is what makes the split-fp16 arithmetic (one scale per tensor) see the same network whatever
    a checkpoint's per-channel scales; (ii) the packed network is the checkpoint's function (torch CPU fp32 on the folded,
    canonicalised convolutions against the oracle's conv + batch_norm pipeline); (iii) every block's inner channel has its
    largest conv1-row magnitude in [1, 2)."""
    from oracle import ref_model
    sd = synth.synthetic_state_dict(0)
    base = _native.canonical_backbone_weights(sd)
    for inner, trunk in ((True, False), (False, True), (True, True)):
        other = _native.canonical_backbone_weights(synth.channel_rescaled_state_dict(sd, spread, seed=spread, inner=inner, trunk=trunk))
        assert np.array_equal(other.view(np.uint32), base.view(np.uint32)), (inner, trunk)
    raw = synth.channel_rescaled_state_dict(sd, spread, seed=1)
    assert max(np.abs(raw[k]).max() / np.abs(sd[k]).max() for k in sd if k.endswith("conv2.weight")) > 2.0 ** (spread - 2)
    stem, blocks, proj = _parse_canonical(base)
    F = torch.nn.functional
    crops = torch.from_numpy(synth.synthetic_crops(2, seed=5))
    x = F.max_pool2d(F.relu(F.conv2d(crops.unsqueeze(1), stem[0], stem[1], 1, 1)), 2, 2)
    for c1, c2, ds, stride in blocks:
        m = torch.maximum(c1[0].abs().amax(dim=(1, 2, 3)), c1[1].abs())
        assert (m >= 1).all() and (m < 2).all()
        h = F.relu(F.conv2d(x, c1[0], c1[1], stride, 1))
        h = F.conv2d(h, c2[0], c2[1], 1, 1)
        x = F.relu(h + (F.conv2d(x, ds[0], ds[1], stride) if ds is not None else x))
    got = F.conv2d(x, proj[0], proj[1])
    want = ref_model.backbone(ref_model.to_torch_state_dict(sd), crops)
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
