"""GPU parity of the torch_data batch path: ut_resample_homography and
ut_gen_crop_matrices through the C ABI against the reference-generated goldens (tests/golden/torch_data.npz,
produced by oracle/gen_goldens.py from the reference's lib.batched_dataset.data_transform) and the oracle, plus the
host mirror (prepare_inputs_targets -> unpack_batched_data -> model -> FK) against the oracle's restatement.

Tolerances.  Resampler: bit-exact given the same resample matrix (float64 arithmetic in the reference's order).
Crop matrices: the reference's look-at chain is six LAPACK inverses and several GEMMs in the dtype of its inputs.  On
the dataset's float32 arrays that is OpenBLAS sgesv / sgemm, whose operation order - hence the last bits - depends on
the kernel OpenBLAS's DYNAMIC_ARCH build picks for the host CPU (numpy 2.2 bundles OpenBLAS 0.3.29; 48 candidate
operation orders of a 4x4 LU solve were tried against it without reproducing its bits), so the reference's own float32
output is not a machine-independent bit pattern.  The fixture therefore also holds the output of the SAME reference
functions on the same values held as float64 (`*_f64chain`, rounded to float32 once at the end): the kernel computes
in float64 in the reference's operation order and must reproduce THOSE values (<= 2 float32 ulps: dgesv's
operation order differs from the kernel's cofactor inverse in the last float64 bits), and must sit within the reference's float32 rounding noise of its float32 output (measured noise
between the two reference runs: 6e-8 on extrinsics, 6e-5 on intrinsics, 3e-5 source pixels on the homography)."""
import numpy as np
import pytest
import torch

from absolutetrack_amd import _native, arch, pipeline, synth, torch_data as td
from absolutetrack_amd.hand import HandModel
from oracle import ref_fk, ref_model, ref_torch_data as rt, scenarios

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def golden(golden_dir):
    return dict(np.load(f"{golden_dir}/torch_data.npz"))


def _dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dt is None else t.to(dt)


@pytest.mark.parametrize("hand", [0, 1])
@pytest.mark.parametrize("src_dtype", ["u8", "f32"])
def test_resampler_is_bit_exact_given_the_reference_matrices(golden, hand, src_dtype):
    c = scenarios.torch_data_case(hand)
    src = c["images"].reshape(-1, *c["images"].shape[2:])
    src = _dev(src) if src_dtype == "u8" else _dev(src.astype(np.float32))
    out = _native.resample_homography(src, _dev(golden[f"h{hand}.resample_xf"].reshape(-1, 4, 4)), (96, 96))
    want = golden[f"h{hand}.images"].reshape(-1, 96, 96)
    assert np.array_equal(out.cpu().numpy(), want)


def _ulps(a, b):
    """Distance in float32 representation steps."""
    ia = a.astype(np.float32).view(np.int32).astype(np.int64)
    ib = b.astype(np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2 ** 31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2 ** 31) - ib, ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("hand", [0, 1])
def test_crop_matrices_match_the_reference(golden, hand):
    c = scenarios.torch_data_case(hand)
    f = c["images"].shape[0]
    m = _native.gen_crop_matrices(_dev(c["extrinsics"]), _dev(c["intrinsics"]), _dev(c["crop_points"]),
                                  torch.full((f,), hand, dtype=torch.int64, device=DEV))
    assert int(m["status"].abs().sum()) == 0
    key = f"h{hand}."
    ext, k, res = (m[n].cpu().numpy() for n in ("extrinsics_xf", "new_intrinsics", "resample_xf"))
    # (1) the reference's functions evaluated on float64 copies of the same inputs: a few float32 ulps (entries
    #     that are zeros / ones in exact arithmetic are compared absolutely)
    for got, name in ((ext, "extrinsics_xf"), (k, "intrinsics"), (res, "resample_xf")):
        want = golden[key + name + "_f64chain"]
        big = np.abs(want) > 1e-3
        u = _ulps(got, want)[big]
        print(name, "ulps vs f64 chain: max", u.max(), "equal", (u == 0).mean())
        assert u.max() <= 2 and (u == 0).mean() > 0.9, (name, u.max(), (u == 0).mean())
        assert np.abs(got - want)[~big].max() < 1e-7, name
    # (2) the reference's float32 run: inside its own rounding noise
    np.testing.assert_allclose(k, golden[key + "intrinsics"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ext, golden[key + "extrinsics_xf"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(res, golden[key + "resample_xf"], rtol=0, atol=6e-5)
    assert (res[:, :, 3] == np.array([0, 0, 0, 1], np.float32)).all()


@pytest.mark.parametrize("hand", [0, 1])
def test_perspective_crop_images_end_to_end(golden, hand):
    """The mirror of _perspective_crop_images (matrices + resampler on the GPU).  The crops differ from the
    reference's float32 run only through the last bits of the matrix chain (<= 6e-5 source pixels, see the module
    docstring): on an image with 30 % white noise that is a few 1e-5 of the [0,1] range, and a sample can flip
    between 'inside' and 'outside' only within that distance of the source border.  With the reference's own matrices
    the crops are bit-equal (test_resampler_is_bit_exact_given_the_reference_matrices)."""
    c = scenarios.torch_data_case(hand)
    img, ext, k = td._perspective_crop_images(c["images"], c["extrinsics"], c["intrinsics"], c["crop_points"], hand,
                                              (96, 96))
    want = golden[f"h{hand}.images"]
    assert img.shape == want.shape and img.dtype == np.float32
    flipped = (img > 0) != (want > 0)
    d = np.abs(img - want)[~flipped]
    print("crop parity hand", hand, "flipped", flipped.sum(), "max", d.max(), "mean", d.mean(), "equal", (d == 0).mean())
    assert flipped.sum() <= 2
    assert d.max() < 2e-4 and d.mean() < 2e-6, (d.max(), d.mean())
    np.testing.assert_allclose(k, golden[f"h{hand}.intrinsics"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ext, golden[f"h{hand}.extrinsics_xf"], atol=2e-7)
    # the crops of the reference resampler driven with the float64-chain matrices: bit-equal
    src = c["images"].reshape(-1, *c["images"].shape[2:]).astype(np.float32)
    want64 = rt.resample_images_batched(src, (96, 96), golden[f"h{hand}.resample_xf_f64chain"].reshape(-1, 4, 4)) / 255
    got64 = _native.resample_homography(_dev(src), _dev(golden[f"h{hand}.resample_xf_f64chain"].reshape(-1, 4, 4)), (96, 96))
    assert np.array_equal(got64.cpu().numpy(), want64.astype(np.float32))


def test_resampler_edge_cases():
    rng = np.random.default_rng(0)
    src = torch.from_numpy(rng.integers(0, 256, (3, 40, 56), dtype=np.uint8)).to(DEV)
    ident = np.eye(4, dtype=np.float32)
    behind = ident.copy()
    behind[2, 2] = 0.0
    behind[2, 3] = -1.0                        # z = -1 everywhere: x/z, y/z negative or zero -> outside
    zero_z = ident.copy()
    zero_z[2, 2] = 0.0                         # z = 0: inf / nan coordinates -> outside, no fault
    xf = torch.from_numpy(np.stack([ident, behind, zero_z])).to(DEV)
    out = _native.resample_homography(src, xf, (40, 56)).cpu().numpy()
    s = src.cpu().numpy().astype(np.float32)
    # identity: integer positions sample exactly; the last row and column have no 2x2 neighbourhood -> 0
    assert np.array_equal(out[0, :39, :55], s[0, :39, :55] / np.float32(255))
    assert (out[0, 39] == 0).all() and (out[0, :, 55] == 0).all()
    assert (out[1, 1:, 1:] == 0).all() and (out[2] == 0).all()
    # oracle agreement on random homographies, including out-of-range ones
    mats = np.tile(ident, (3, 1, 1))
    mats[:, :2, :3] += rng.normal(0, 0.2, (3, 2, 3)).astype(np.float32)
    mats[:, 2, :2] += rng.normal(0, 2e-3, (3, 2)).astype(np.float32)
    mats[:, :2, 3] = rng.normal(0, 5, (3, 2)).astype(np.float32)
    got = _native.resample_homography(src, torch.from_numpy(mats).to(DEV), (24, 32)).cpu().numpy()
    want = rt.resample_images_batched(s, (24, 32), mats) / 255
    assert np.array_equal(got, want.astype(np.float32))
    # empty batch, argument checks, no CPU fallback
    assert _native.resample_homography(src[:0], xf[:0], (8, 8)).shape == (0, 8, 8)
    with pytest.raises(ValueError):
        _native.resample_homography(src, xf[:2], (8, 8))
    with pytest.raises(ValueError):
        _native.resample_homography(src.double(), xf, (8, 8))
    with pytest.raises(_native.NativeLibraryError):
        _native.resample_homography(src.cpu(), xf.cpu(), (8, 8))
    lib = _native.load_library()
    assert lib.ut_resample_homography(None, None, 0, 1, 4, 4, None, 4, 4, None, None) != 0
    assert b"ut_resample_homography" in lib.ut_last_error(None)


def test_unbuildable_crop_raises_like_the_reference():
    c = scenarios.torch_data_case(0)
    pts = c["crop_points"].copy()
    # one enclosing point behind the first view's camera: z < 1e-4 in the crop camera -> ValueError (crop.py:25-26)
    c2w = np.linalg.inv(c["extrinsics"][0, 0].astype(np.float64))
    pts[0, 0] = (c2w[:3, 3] - 0.05 * c2w[:3, 2]).astype(np.float32)
    with pytest.raises(ValueError, match="Unable to create crop camera"):
        td._perspective_crop_images(c["images"], c["extrinsics"], c["intrinsics"], pts, 0, (96, 96))
    with pytest.raises(ValueError):
        rt.gen_crop_matrices(c["extrinsics"][0], c["intrinsics"][0], pts[0], False, (96, 96))
    with pytest.raises(ValueError):
        td._perspective_crop_images(c["images"], c["extrinsics"], c["intrinsics"], c["crop_points"], 0, (96, 64))


def _raw_sample(hand: int, lab) -> td.RawSample:
    """A RawSample in the on-disk units (mm) from the metre-valued scenario; the label pose is the target."""
    c = scenarios.torch_data_case(hand)
    f = c["images"].shape[0]
    frames = [5 + 40 * i for i in range(f)]
    ext_mm = c["extrinsics"].copy()
    ext_mm[..., :3, 3] *= 1000.0
    hm = {k[3:]: torch.from_numpy(v) for k, v in lab.items() if k.startswith("hm.")}
    z = torch.zeros(22)
    model = HandModel(joint_rotation_axes=hm["joint_rotation_axes"], joint_rest_positions=hm["joint_rest_positions"],
                      joint_frame_index=z, joint_parent=z, joint_first_child=z, joint_next_sibling=z,
                      landmark_rest_positions=hm["landmark_rest_positions"],
                      landmark_rest_bone_weights=hm["landmark_rest_bone_weights"],
                      landmark_rest_bone_indices=hm["landmark_rest_bone_indices"], hand_scale=torch.tensor(1.0))
    wrist = lab["wrist_transforms"][frames, hand].astype(np.float32)
    ja = lab["joint_angles"][frames, hand].astype(np.float32)
    return td.RawSample(images=c["images"], extrinsics=ext_mm, intrinsics=c["intrinsics"].copy(),
                        enclosing_points=c["crop_points"] * np.float32(1000.0), hand=np.full(f, hand, np.float32),
                        hand_model=model, wrist=wrist.copy(), joint_angles=ja, solved_wrist_xfs=wrist.copy(),
                        solved_joint_angles=ja.copy(), generic_hand_model=model, pinch=np.zeros(f, np.float32))


@pytest.mark.parametrize("use_skel", [True, False])
def test_sequence_batch_through_the_model_matches_oracle(use_skel):
    """prepare_inputs_targets -> collate -> unpack_batched_data -> model (temporal memory engaged from step 1) ->
    skin_landmarks, as run_inference_torch_data.py:88-130 drives it, against the oracle fed with the same crops."""
    from absolutetrack_amd import bundles
    from absolutetrack_amd.model import UmeTrackModel
    lab = pipeline.load_labels()
    pairs = [td.prepare_inputs_targets(_raw_sample(h, lab), (96, 96)) for h in (0, 1)]
    model_input = bundles.collate([p[0] for p in pairs])
    model_target = bundles.collate([p[1] for p in pairs])
    assert tuple(model_input.left_images.shape) == (2, 4, 2, 96, 96)
    sd = synth.synthetic_state_dict(0)
    model = UmeTrackModel(sd)
    model.eval()
    model.to(DEV)
    gt_kp, out_kp = td.eval_batch_keypoints(model, model_input, model_target, "multiv", use_skel, DEV)
    err_mm = td._eval_batch(model, model_input, model_target, "multiv", use_skel, DEV)
    assert tuple(err_mm.shape) == (2,) and torch.allclose(err_mm, (gt_kp - out_kp).norm(dim=-1).mean(dim=(1, 2)) * 1000, rtol=1e-4)
    assert tuple(out_kp.shape) == (2, 4, 21, 3) and tuple(gt_kp.shape) == (2, 4, 21, 3)
    # oracle: same crops and matrices, reference arithmetic on the CPU
    om = ref_model.OracleModel(sd)
    hm_np = {k[3:]: v for k, v in lab.items() if k.startswith("hm.")}
    li, k, x = (t.numpy() for t in (model_input.left_images, model_input.intrinsics, model_input.extrinsics_xf))
    hm_left = model_input.orig_pose_data.left_hand_model
    steps = rt.unpack_batched_data(li, k, x, model_input.hand_idx.numpy(), hm_left.joint_rotation_axes.numpy(),
                                   hm_left.joint_rest_positions.numpy(), "multiv")
    worst = 0.0
    for i, s in enumerate(steps):
        o = om.forward(torch.from_numpy(s["images"]), torch.from_numpy(s["intrinsics"]), torch.from_numpy(s["extrinsics"]),
                       torch.from_numpy(s["sample_range"]), torch.from_numpy(s["memory_idx"]),
                       torch.from_numpy(s["use_memory"]), torch.from_numpy(s["hand_idx"]),
                       torch.from_numpy(s["axes"]), torch.from_numpy(s["rest"]), known_skeleton=use_skel)
        # eval_batch skins with mirrored_hand_model(left model, hand == 1): for the right-hand sequence that undoes the
        # mirroring prepare_inputs_targets applied, i.e. both sequences are skinned with the sample's own model (m)
        hmb = dict(hm_np)
        hmb["joint_rest_positions"] = hm_np["joint_rest_positions"] * np.float32(0.001)
        hmb["landmark_rest_positions"] = hm_np["landmark_rest_positions"] * np.float32(0.001)
        for b in range(2):
            kp = ref_fk.skin_landmarks(hmb, o["joint_angles"][b].numpy(), o["wrist_xfs"][b].numpy())
            worst = max(worst, float(np.abs(kp - out_kp[b, i].numpy()).max()))
    assert worst < 1e-6, worst      # metres: north_star's 1e-3 mm keypoint tolerance
    # targets: FK of the label pose reproduces the enclosing points' first 21 landmarks (metres)
    c0 = scenarios.torch_data_case(0)
    np.testing.assert_allclose(gt_kp[0].numpy(), c0["crop_points"][:, :21], atol=5e-6)
