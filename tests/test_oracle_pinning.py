"""CPU tests: the oracle restatement against the golden vectors produced by the
reference's own code (oracle/gen_goldens.py) and against the reference's stored
evaluation outputs.  These pin the oracle; the GPU parity tests then compare the HIP
path with the oracle."""
import os

import numpy as np
import pytest
import torch

from absolutetrack_amd import arch, synth
from oracle import ref_camera, ref_fk, ref_model, scenarios


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_state_dict_schema():
    spec = arch.state_dict_spec()
    assert len(spec) == 252
    n_float = sum(int(np.prod(s)) for _k, s, kind in spec if kind not in ("bn_nbt", "bn_mean", "bn_var"))
    assert n_float == 4_251_227
    assert sum(1 for _k, _s, kind in spec if kind == "bn_nbt") == 39
    sd = synth.synthetic_state_dict(0)
    assert list(sd) == [k for k, _s, _kind in spec]
    # counter based: regenerating gives identical bits
    sd2 = synth.synthetic_state_dict(0)
    assert all(np.array_equal(sd[k], sd2[k]) for k in sd)


@pytest.mark.parametrize("known", [True, False])
def test_model_oracle_matches_reference_goldens(golden_dir, known):
    g = _load(golden_dir, "model_known.npz" if known else "model_unknown.npz")
    m = ref_model.OracleModel(synth.synthetic_state_dict(0))
    axes, rest = (torch.from_numpy(a) for a in scenarios.skeleton_m())
    for si, st in enumerate(scenarios.model_steps(known)):
        t = {k: torch.from_numpy(v) for k, v in st.items()}
        taps = {}
        o = m.forward(t["images"], t["intrinsics"], t["extrinsics"], t["sample_range"], t["memory_idx"],
                      t["use_memory"], t["hand_idx"], axes, rest, known_skeleton=known, taps=taps)
        p = f"s{si}."
        np.testing.assert_allclose(taps["proj"].numpy(), g[p + "proj"], atol=2e-6)
        np.testing.assert_allclose(o["raw"].numpy(), g[p + "raw"], atol=5e-6)
        np.testing.assert_allclose(o["joint_angles"].numpy(), g[p + "joint_angles"], atol=5e-6)
        np.testing.assert_allclose(o["wrist_xfs"].numpy(), g[p + "wrist_xfs"], atol=5e-6)
        np.testing.assert_allclose(o["landmark_uncertainty_sigmas"].numpy(), g[p + "sigmas"], atol=5e-6)
        np.testing.assert_allclose(m.temporal.mem.numpy(), g[p + "mem_state"], atol=2e-6)
        np.testing.assert_allclose(m.temporal.prev_ext.numpy(), g[p + "prev_ext_state"], atol=0)
        if not known:
            np.testing.assert_allclose(o["skel_scales"].numpy(), g[p + "skel_scales"], atol=5e-6)
        if si == 0:
            np.testing.assert_allclose(taps["stem"][:, :, ::6, ::6].numpy(), g["s0.stem_sub"], atol=1e-6)
            np.testing.assert_allclose(taps["_layers.1.1"][:, :, ::6, ::6].numpy(), g["s0.layer1_sub"], atol=1e-6)
            np.testing.assert_allclose(taps["_layers.2.2"][:, :, ::3, ::3].numpy(), g["s0.layer2_sub"], atol=1e-6)
            np.testing.assert_allclose(taps["_layers.3.4"][:, ::2, ::2, ::2].numpy(), g["s0.layer3_sub"], atol=1e-6)
            np.testing.assert_allclose(taps["_layers.4.1"][:, ::4].numpy(), g["s0.layer4_sub"], atol=1e-6)


def test_single_view_rejected_in_unknown_mode():
    m = ref_model.OracleModel(synth.synthetic_state_dict(0))
    st = scenarios.model_steps(True)[0]      # has a single-view sample
    t = {k: torch.from_numpy(v) for k, v in st.items()}
    with pytest.raises(AssertionError):
        m.forward(t["images"], t["intrinsics"], t["extrinsics"], t["sample_range"], t["memory_idx"],
                  t["use_memory"], t["hand_idx"], known_skeleton=False)


def test_fk_oracle_matches_stored_reference_keypoints(golden_dir):
    """gt_keypoints in sample_data/user05/*.npy are landmarks_from_hand_pose(label model, label pose)
    computed by the reference with pytorch3d (run_eval_known_skeleton.py:87-89)."""
    g = _load(golden_dir, "fk_user05.npz")
    n_checked = 0
    for rec in ("00", "02", "11"):
        p = f"r{rec}."
        hm = {k[len(p) + 3:]: g[k] for k in g.files if k.startswith(p + "hm.")}
        for hand in (0, 1):
            valid = g[p + "valid_tracking"][hand]
            lm = ref_camera.landmarks_from_pose(hm, g[p + "joint_angles"][:, hand][0], g[p + "wrist_transforms"][0, hand], hand)
            assert lm.shape == (21, 3)
            xf = g[p + "wrist_transforms"][:, hand].copy()
            if hand == 1:
                xf[:, :, 0] *= -1
            lm = ref_fk.skin_landmarks(hm, g[p + "joint_angles"][:, hand].astype(np.float32), xf.astype(np.float32))
            err = np.abs(lm - g[p + "gt_keypoints"][hand])[valid]
            assert err.max() < 1e-3, (rec, hand, err.max())          # mm
            n_checked += int(valid.sum())
    assert n_checked > 250


def test_fk_leading_dims():
    hm = scenarios.hand_model_mm()
    lab = scenarios.labels()
    ja = lab["joint_angles"][:6].astype(np.float32)                    # [6,2,22]
    xf = lab["wrist_transforms"][:6].astype(np.float32)
    full = ref_fk.skin_landmarks(hm, ja, xf)
    assert full.shape == (6, 2, 21, 3)
    one = ref_fk.skin_landmarks(hm, ja[3, 1], xf[3, 1])
    np.testing.assert_allclose(full[3, 1], one, atol=1e-5)


def test_geometry_oracle_matches_reference_goldens(golden_dir):
    g = _load(golden_dir, "geometry_rec00.npz")
    lab = scenarios.labels()
    hm = scenarios.hand_model_mm()
    n_maps = 0
    for fi in g["frames"]:
        cams = [ref_camera.camera_from_json(
            dict(zip(("ImageSizeX", "ImageSizeY", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "k5", "k6"),
                     lab["cameras"][ci])) | {"DistortionModel": "FishEye62"},
            lab["camera_to_world_transforms"][fi, ci]) for ci in range(4)]
        for hand in (0, 1):
            key = f"f{fi}.h{hand}."
            crops = ref_camera.gen_crop_cameras(cams, lab["camera_angles"], hm, lab["joint_angles"][fi, hand],
                                                lab["wrist_transforms"][fi, hand], hand)
            assert list(crops) == list(g[key + "cams"])
            for ci, cc in crops.items():
                ck = key + f"c{ci}."
                np.testing.assert_allclose(cc["f"], g[ck + "f"], rtol=1e-9)
                np.testing.assert_allclose(cc["c"], g[ck + "c"], rtol=0)
                np.testing.assert_allclose(cc["T"], g[ck + "T"], atol=1e-9)
                m = ref_camera.warp_map(cams[ci], cc)
                np.testing.assert_allclose(m[::4, ::4], g[ck + "map_sub"], atol=2e-4)   # px, f32 cast
                k, ext = ref_camera.network_inputs_for_crop(cc)
                np.testing.assert_allclose(k, g[ck + "K"], rtol=1e-6)
                n_maps += 1
    assert n_maps == 40


def test_remap_modes_self_consistent():
    """cv2.remap is absent: float-bilinear and the OpenCV fixed-point emulation must agree to within
    the quantisation the latter introduces (PARITY UNPINNED against OpenCV itself)."""
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (60, 80), dtype=np.uint8)
    m = np.stack([rng.uniform(-3, 83, (40, 40)), rng.uniform(-3, 63, (40, 40))], -1).astype(np.float32)
    a = ref_camera.remap_bilinear(src, m, "float")
    b = ref_camera.remap_bilinear(src, m, "cv2").astype(np.float32)
    assert np.abs(a - b).max() <= 255 * (2 / 32) + 0.5 + 1e-3
    # integer coordinates sample exactly, out-of-range gives the constant border 0
    mi = np.stack(np.meshgrid(np.arange(80), np.arange(60)), -1).astype(np.float32)
    assert np.array_equal(ref_camera.remap_bilinear(src, mi, "cv2"), src)
    mo = np.full((2, 2, 2), -1, np.float32)
    assert ref_camera.remap_bilinear(src, mo, "cv2").max() == 0
    assert ref_camera.cv2_bilinear_tab().sum(-1).min() == 32768 == ref_camera.cv2_bilinear_tab().sum(-1).max()


def test_torch_data_oracle_matches_reference_goldens(golden_dir):
    """Row f2: the restatement of lib/batched_dataset/data_transform.py is bit-identical to the reference's own
    functions on the seeded sequences (crops, crop extrinsics/intrinsics, resample matrices)."""
    from oracle import ref_torch_data as rt
    g = _load(golden_dir, "torch_data.npz")
    for hand in (0, 1):
        c = scenarios.torch_data_case(hand)
        img, ext, intr = rt.perspective_crop_images(c["images"], c["extrinsics"], c["intrinsics"], c["crop_points"], hand,
                                                    (96, 96))
        key = f"h{hand}."
        assert np.array_equal(img, g[key + "images"])
        assert np.array_equal(ext, g[key + "extrinsics_xf"])
        assert np.array_equal(intr, g[key + "intrinsics"])
        res = np.stack([rt.gen_crop_matrices(c["extrinsics"][f], c["intrinsics"][f], c["crop_points"][f], hand == 1,
                                             (96, 96))[2] for f in range(c["images"].shape[0])])
        assert np.array_equal(res, g[key + "resample_xf"])
        # the scenario exercises both fully covered crops and crops that leave the source image
        cover = (g[key + "images"] > 0).mean(axis=(2, 3))
        assert cover.max() == 1.0 and cover.min() < 0.95
