"""GPU parity tests (run with -m gpu on the MI355X box): every C-ABI entry point of
libumetrack_hip.so against the CPU oracle and the committed reference goldens, on the same
seeded inputs.  Tolerances are BASELINE.json's: 1e-4 rad on joint angles, 1e-3 mm on 3D
keypoints / translations (the network works in metres: 1e-6 m)."""
import os

import numpy as np
import pytest
import torch

from absolutetrack_amd import _native, arch, synth
from oracle import ref_camera, ref_fk, ref_model, scenarios

pytestmark = pytest.mark.gpu

ANGLE_TOL = 1e-4        # rad
METRE_TOL = 1e-6        # 1e-3 mm
DEV = "cuda:0"


@pytest.fixture(scope="module")
def engine():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    yield eng
    eng.close()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_library_is_the_in_tree_build(engine):
    assert os.path.samefile(engine.lib._name, _native.LIB_PATH)


def test_load_pretrained_model_file_to_engine(engine, tmp_path):
    """File -> load_pretrained_model -> .to('cuda') -> native handle (lib/models/model_loader.py:53-88,
    lib/tracker/tracker.py:97-98): same features, bit for bit, as the handle built from the in-memory weights."""
    from lib.models.model_loader import load_pretrained_model
    from lib.models.umetrack_model import InputFrameData, InputFrameDesc
    path = str(tmp_path / "w.torch")
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synthetic_state_dict(0).items()}, path)
    m = load_pretrained_model(path)
    m.eval()
    m.to("cuda")
    crops = _dev(synth.synthetic_crops(4, seed=8))
    assert torch.equal(m.engine.backbone(crops), engine.backbone(crops))
    fd = InputFrameData(crops, torch.tensor([[130.0, 0, 47.5], [0, 130.0, 47.5], [0, 0, 1]]).repeat(4, 1, 1),
                        torch.eye(4).repeat(4, 1, 1))
    desc = InputFrameDesc(torch.tensor([[0, 2], [2, 4]]), torch.tensor([0, 1]), torch.tensor([False, False]),
                          torch.tensor([0, 1]))
    out = m.regress_pose_pred_skel_scale(fd, desc)
    assert out.joint_angles.shape == (2, 22) and out.skel_scales.shape == (2,) and out.wrist_xfs.device.type == "cuda"


def test_backbone_matches_oracle(engine):
    crops = synth.synthetic_crops(7, seed=3)
    taps = {}
    want = ref_model.backbone(ref_model.to_torch_state_dict(synth.synthetic_state_dict(0)), torch.from_numpy(crops), taps)
    got = engine.backbone(_dev(crops)).cpu()
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert err < 2e-5 * max(1.0, scale), (err, scale)
    # batch-size independence: chunked passes give identical bits
    engine.set_backbone_chunk(3)
    got2 = engine.backbone(_dev(crops)).cpu()
    engine.set_backbone_chunk(0)
    assert torch.equal(got, got2)


def test_backbone_edge_batches(engine):
    assert engine.backbone(torch.empty(0, 96, 96, device=DEV)).shape == (0, 72, 6, 6)
    one = synth.synthetic_crops(1, seed=9)
    want = ref_model.backbone(ref_model.to_torch_state_dict(synth.synthetic_state_dict(0)), torch.from_numpy(one))
    got = engine.backbone(_dev(one)).cpu()
    assert (got - want).abs().max().item() < 2e-5
    with pytest.raises(ValueError):
        engine.backbone(torch.zeros(2, 64, 64, device=DEV))


@pytest.fixture(scope="module")
def split_engine():
    """The same weights with the eligible backbone convolutions on the split-fp16 kernel (conv_split.hip) for EVERY launch
    size, so that the small parity cases below go through it."""
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    eng.set_conv_arithmetic("split_f16_always")
    yield eng
    eng.close()


def test_split_f16_backbone_matches_oracle(engine, split_engine):
    """conv_split.hip (fp16 matrix cores, two-piece operand splits, 3 products) against the fp32 oracle at the fp32
    kernels' tolerance, on a ragged launch (7 crops: 1008 pixels at 12x12 = three full 256-row tiles and one of 240; 252
    pixels at 6x6 = one partial tile in two column tiles) - and its distance to the fp32-MFMA mode."""
    crops = synth.synthetic_crops(7, seed=3)
    want = ref_model.backbone(ref_model.to_torch_state_dict(synth.synthetic_state_dict(0)), torch.from_numpy(crops))
    got = split_engine.backbone(_dev(crops)).cpu()
    scale = max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() < 2e-5 * scale
    fp32 = engine.backbone(_dev(crops)).cpu()
    assert not torch.equal(got, fp32)                       # a different kernel did run
    assert (got - fp32).abs().max().item() < 1e-5 * scale
    assert torch.equal(split_engine.backbone(_dev(crops)).cpu(), got)       # deterministic


@pytest.mark.parametrize("n_crops", [1, 5, 37, 300])
def test_split_f16_layer2_kernels_against_the_chunked_kernel(engine, split_engine, n_crops):
    """Layer2's five stride-1 64 -> 64 convolutions through conv_w4.hip (the default: one 24x24 map x 64 channels per tile, 16-channel
    slices - the chunked kernel's products summed slice-half by slice-half) and through conv_c64k.hip (weights resident in registers,
    K split across the two waves of a SIMD) against conv_split_kernel<256, 64, 8, 1, true> on the same tensors: fp32 rounding apart,
    far inside the split arithmetic's own distance to fp32, and deterministic.  1 crop = one tile (fewer tiles than workgroups), 300
    crops = 300 tiles on 256 workgroups (the tile queue hands out second tiles; the three patch buffers wrap across tiles)."""
    crops = _dev(synth.synthetic_crops(n_crops, seed=23 + n_crops))
    w4 = split_engine.backbone(crops)
    try:
        split_engine.set_resident_weights(5)             # layer3 / layer4 as in the default, layer2 through the chunked kernel
        chunked = split_engine.backbone(crops)
        split_engine.set_resident_weights(4)             # ... layer2 through conv_c64k
        pair = split_engine.backbone(crops)
    finally:
        split_engine.set_resident_weights(1)
    assert torch.isfinite(w4).all() and torch.isfinite(pair).all()
    assert torch.equal(split_engine.backbone(crops), w4)             # deterministic
    fp32 = engine.backbone(crops)
    scale = max(1.0, fp32.abs().max().item())
    assert (w4 - chunked).abs().max().item() < 2e-6 * scale
    assert (pair - chunked).abs().max().item() < 2e-6 * scale
    assert (w4 - fp32).abs().max().item() < 1e-5 * scale


@pytest.mark.parametrize("n_crops", [1, 2, 5, 37, 300])
def test_split_f16_four_wave_kernel_has_the_chunked_kernel_bits(engine, split_engine, n_crops):
    """conv_w4.hip on layer3's 128 -> 128 and layer4's 256 -> 256 convolutions (four waves, each all 288 pixels of a tile x 32 output
    channels; weights global -> registers, the patch split on its way into LDS at padded image coordinates, one barrier per slice)
    against conv_split_kernel<256, 128, 4, 2, true>: per output element the same products in the same order, so the backbone's
    features are equal bit for bit (layer2 through the chunked kernel on both sides: ut_set_resident_weights 5 against 0).  A tile is
    2 whole 12x12 maps / 8 whole 6x6 maps: 1 crop = half a tile at 12x12 and an eighth at 6x6 (pixels beyond the tensor inside the only
    tile), 2 crops = one exact tile at 12x12, 5 and 37 crops = ragged last tiles at both sizes, 300 crops = 150 + 2 x 38 tiles on 256
    workgroups (the tile queue hands out second tiles at 6x6)."""
    crops = _dev(synth.synthetic_crops(n_crops, seed=51 + n_crops))
    try:
        split_engine.set_resident_weights(5)
        got = split_engine.backbone(crops)
        again = split_engine.backbone(crops)
        split_engine.set_resident_weights(0)
        chunked = split_engine.backbone(crops)
    finally:
        split_engine.set_resident_weights(1)
    assert torch.isfinite(got).all()
    assert torch.equal(got, chunked)
    assert torch.equal(again, got)                                   # deterministic


@pytest.mark.parametrize("n_crops", [1, 2, 5, 37, 300])
def test_split_f16_phase_plane_entries_against_the_gather_kernel(engine, split_engine, n_crops):
    """The stride-2 3x3 entries of layer3 (64 -> 128, 24x24 -> 12x12) and layer4 (128 -> 256, 12x12 -> 6x6) through conv_w4.hip's
    phase-plane form (the default: the input's four (row parity, column parity) planes at padded coordinates of the OUTPUT map, two
    16-channel planes per patch buffer, products summed plane by plane) against conv_split_kernel<256, 128, 4, 2, false> (gathers
    tap by tap) on the same tensors (ut_set_resident_weights 1 against 6): the same products in another order - fp32 rounding
    apart, far inside the split arithmetic's own distance to fp32 - and deterministic.  1 crop = half / an eighth of a tile (pixels
    beyond the tensor inside the only tile, and a next tile that does not exist), 2 crops = one exact tile at 12x12, 5 and 37 =
    ragged last tiles, 300 = 150 + 2 x 38 tiles on 256 workgroups (second tiles from the queue)."""
    crops = _dev(synth.synthetic_crops(n_crops, seed=77 + n_crops))
    planes = split_engine.backbone(crops)
    try:
        split_engine.set_resident_weights(6)
        gather = split_engine.backbone(crops)
    finally:
        split_engine.set_resident_weights(1)
    assert torch.isfinite(planes).all()
    assert torch.equal(split_engine.backbone(crops), planes)         # deterministic
    assert not torch.equal(planes, gather)                           # a different kernel did run
    fp32 = engine.backbone(crops)
    scale = max(1.0, fp32.abs().max().item())
    assert (planes - gather).abs().max().item() < 2e-6 * scale
    assert (planes - fp32).abs().max().item() < 1e-5 * scale
    split_engine.poll_status()


def test_split_f16_fused_layer1_blocks_match_the_two_launch_form(engine, split_engine):
    """conv_block32.hip (layer1's BasicBlocks as one launch each, the intermediate in LDS) against the same arithmetic as two
    convolution launches per block: the forms differ only in the intermediate's power-of-two scale (a bound there, the
    measured maximum here), i.e. in how the smallest values round - far inside the split arithmetic's own distance to fp32.
    37 crops = 444 tiles on 256 persistent workgroups (the tile queue and the double-buffered patch are exercised).  The same
    switch turns off conv_c32s2.hip (layer2's stride-2 3x3 and its 1x1 shortcut as one launch: 222 tiles of 4 output rows here;
    without it the shortcut is an fp32-instruction launch), so both fused kernels are held against their separate-launch forms."""
    crops = _dev(synth.synthetic_crops(37, seed=11))
    fused = split_engine.backbone(crops)
    split_engine.set_block_fusion(False)
    try:
        two = split_engine.backbone(crops)
    finally:
        split_engine.set_block_fusion(True)
    assert torch.equal(split_engine.backbone(crops), fused)          # deterministic
    fp32 = engine.backbone(crops)
    scale = max(1.0, fp32.abs().max().item())
    assert (fused - two).abs().max().item() < 2e-6 * scale
    assert (fused - fp32).abs().max().item() < 1e-5 * scale
    assert (two - fp32).abs().max().item() < 1e-5 * scale
    want = ref_model.backbone(ref_model.to_torch_state_dict(synth.synthetic_state_dict(0)), crops[:6].cpu())
    assert (fused[:6].cpu() - want).abs().max().item() < 2e-5 * scale


def _rescaled_state_dict(k: int):
    """synthetic_state_dict(0) with every backbone block's bn1.{weight,bias} x 2^-k and conv2.weight x 2^k: the activations
    between conv1 and conv2 of every block are 2^-k of the original network's, everything else - in exact arithmetic and
    in fp32, where powers of two commute with every rounding and with ReLU - is unchanged."""
    sd = dict(synth.synthetic_state_dict(0))
    down, up = np.float32(2.0 ** -k), np.float32(2.0 ** k)
    n = 0
    for key in list(sd):
        if "_image_backbone.0._layers." not in key:
            continue
        if key.endswith(".bn1.weight") or key.endswith(".bn1.bias"):
            sd[key] = sd[key] * down
            n += 1
        elif key.endswith(".conv2.weight"):
            sd[key] = sd[key] * up
            n += 1
    assert n == 36          # 12 blocks x (bn1.weight, bn1.bias, conv2.weight)
    return sd


@pytest.mark.parametrize("k", [8, 12, 14, 16, 40, -8, -14, -16, -40])
def test_split_f16_power_of_two_rescale_invariance(engine, split_engine, k):
    """fp32 `relu(bn(conv))` (lib/models/backbone_resnet.py:56-72) has no range precondition: a network whose inner
    activations are 2^-k (or 2^+k: far beyond fp16's 65504) of another's gives the same bits.  The split-fp16 kernels scale
    every layer's activations by a power of two taken from the producer's max word before the two-piece split, so they
    inherit that: identical features whatever k, and the reference's own golden outputs at the path's tolerances."""
    crops = _dev(synth.synthetic_crops(7, seed=3))
    base32, base_split = engine.backbone(crops), split_engine.backbone(crops)
    eng = _native.HipEngine(_rescaled_state_dict(k), DEV)
    try:
        assert torch.equal(eng.backbone(crops), base32)                    # the fp32 kernels: bit for bit
        eng.set_conv_arithmetic("split_f16_always")
        got = eng.backbone(crops)
        eng.poll_status()                                                  # nothing to report at any k
        scale = max(1.0, base32.abs().max().item())
        assert (got - base32).abs().max().item() < 1e-5 * scale            # split vs fp32: as for the unscaled network
        assert torch.equal(got, base_split)                                # and the same bits as the unscaled split run
    finally:
        eng.close()


@pytest.mark.parametrize("spread", [8, 12, 16, 24])
def test_split_f16_per_channel_rescale_invariance(engine, split_engine, spread):
    """The split arithmetic keeps ONE power-of-two scale per activation tensor and one per weight tensor, so a channel far below
    its tensor's largest (a near-dead BatchNorm channel whose consumer weights compensate: the same fp32 function) would lose its
    second fp16 piece.  ut_create therefore brings every inner and trunk channel to a canonical power-of-two scale before packing
    (exact; csrc/ut_api.hip::fold_backbone).  Here every inner AND trunk channel of the backbone is rescaled by its own random
    2^k, k in [-spread, spread] (channel spread up to 2^48): the fp32 kernels give the same bits (powers of two commute with every
    rounding), the split kernels give the same bits as on the original network (both pack to the same tensors), and the
    reference's own goldens hold at the path's tolerances."""
    crops = _dev(synth.synthetic_crops(7, seed=3))
    base32, base_split = engine.backbone(crops), split_engine.backbone(crops)
    eng = _native.HipEngine(synth.channel_rescaled_state_dict(synth.synthetic_state_dict(0), spread, seed=100 + spread), DEV)
    try:
        assert torch.equal(eng.backbone(crops), base32)
        eng.set_conv_arithmetic("split_f16_always")
        got = eng.backbone(crops)
        eng.poll_status()
        assert (got - base32).abs().max().item() < 1e-5 * max(1.0, base32.abs().max().item())
        assert torch.equal(got, base_split)
    finally:
        eng.close()


@pytest.mark.parametrize("known", [True, False])
def test_split_f16_per_channel_rescaled_network_matches_reference_goldens(golden_dir, known):
    """A per-channel rescaled network (spread 2^+-24) through backbone + head against the reference's own outputs."""
    eng = _native.HipEngine(synth.channel_rescaled_state_dict(synth.synthetic_state_dict(0), 24, seed=7), DEV)
    try:
        eng.set_conv_arithmetic("split_f16_always")
        test_model_matches_reference_goldens(eng, golden_dir, known)
        eng.poll_status()
    finally:
        eng.close()


@pytest.mark.parametrize("known", [True, False])
@pytest.mark.parametrize("k", [14, -16])
def test_split_f16_rescaled_network_matches_reference_goldens(golden_dir, known, k):
    """The rescaled networks above through backbone + head against the reference's own outputs (1e-4 rad / 1e-3 mm)."""
    eng = _native.HipEngine(_rescaled_state_dict(k), DEV)
    try:
        eng.set_conv_arithmetic("split_f16_always")
        test_model_matches_reference_goldens(eng, golden_dir, known)
        eng.poll_status()
    finally:
        eng.close()


def test_split_f16_large_activations_and_non_finite_guard(engine):
    """One convolution's weights x 1e6 puts its block's activations at ~1e6-1e8, far beyond fp16's 65504: the split mode
    scales them into range and agrees with the exact-fp32 mode on the same weights to fp32-level relative error.  What has no
    scale is an infinity or a NaN among a layer's inputs: a sticky device-side flag, raised at the next status read; the
    exact-fp32 mode runs the same weights without complaint (it propagates the infinity like the reference)."""
    sd = dict(synth.synthetic_state_dict(0))
    key = "_feature_extractor._image_backbone.0._layers.2.0.conv2.weight"
    sd[key] = sd[key] * np.float32(1e6)
    crops = _dev(synth.synthetic_crops(6, seed=4))
    eng = _native.HipEngine(sd, DEV)
    try:
        want = eng.backbone(crops)
        eng.set_conv_arithmetic("split_f16_always")
        got = eng.backbone(crops)
        eng.poll_status()
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    finally:
        eng.close()
    sd = dict(synth.synthetic_state_dict(0))
    key = "_feature_extractor._image_backbone.0._layers.2.0.bn2.bias"
    sd[key] = sd[key].copy()
    sd[key][3] = np.float32(np.inf)
    eng = _native.HipEngine(sd, DEV)
    try:
        eng.backbone(crops)
        eng.poll_status()                                   # fp32 mode: nothing to report
        eng.set_conv_arithmetic("split_f16_always")
        eng.backbone(crops)
        with pytest.raises(FloatingPointError, match="infinity or a NaN"):
            eng.poll_status()
        eng.poll_status()                                   # the flag was cleared by the read that reported it
    finally:
        eng.close()


def test_split_f16_large_batch(engine):
    """The default split mode engages on launches that fill the chip: 2048 + 37 crops (ragged last tiles at every
    resolution), against the fp32-MFMA mode on the same crops and the oracle on a few of them."""
    n = 2048 + 37
    g = torch.Generator(device=DEV)
    g.manual_seed(12)
    crops = torch.rand(n, 96, 96, device=DEV, generator=g)
    fp32 = engine.backbone(crops)
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    try:
        eng.set_conv_arithmetic("split_f16")
        got = eng.backbone(crops)
        assert torch.equal(eng.backbone(crops), got)
        few = eng.backbone(crops[:5])                       # too few tiles: the fp32 kernels, bit for bit
        assert torch.equal(few, engine.backbone(crops[:5]))
    finally:
        eng.close()
    scale = max(1.0, fp32.abs().max().item())
    assert not torch.equal(got, fp32)
    assert (got - fp32).abs().max().item() < 1e-5 * scale
    idx = [0, 1, 1000, 2047, n - 1]
    want = ref_model.backbone(ref_model.to_torch_state_dict(synth.synthetic_state_dict(0)), crops[idx].cpu())
    assert (got[idx].cpu() - want).abs().max().item() < 2e-5 * scale


def _run_steps(engine, known, want_raw=True):
    engine.reset_memory()
    axes, rest = scenarios.skeleton_m()
    skel = _dev(np.stack([axes, rest])[None]) if known else None
    outs = []
    for st in scenarios.model_steps(known):
        feat = engine.backbone(_dev(st["images"]))
        sr = st["sample_range"]
        pose, raw = engine.fuse_temporal_regress(
            feat, _dev(st["intrinsics"]), _dev(st["extrinsics"]), _dev(sr), _dev(st["memory_idx"]),
            _dev(st["use_memory"]), _dev(st["hand_idx"]), int(st["memory_idx"].max()) + 1,
            bool(((sr[:, 1] - sr[:, 0]) == 2).all()), skel,
            _native.UT_MODE_KNOWN if known else _native.UT_MODE_UNKNOWN, want_raw=want_raw)
        mem, ext = engine.get_memory()
        outs.append((feat.cpu().numpy(), pose.cpu().numpy(), raw.cpu().numpy(), mem.cpu().numpy(), ext.cpu().numpy()))
    return outs


@pytest.mark.parametrize("known", [True, False])
def test_split_f16_model_matches_reference_goldens(split_engine, golden_dir, known):
    """The reference's own outputs (tests/golden/model_*.npz), same tolerances, with the backbone's and the pose regressor's 3x3
    convolutions on the split-fp16 kernels (calibrated scales, the default)."""
    test_model_matches_reference_goldens(split_engine, golden_dir, known)
    split_engine.poll_status()


@pytest.mark.parametrize("known", [True, False])
def test_split_f16_dynamic_scales_match_reference_goldens(golden_dir, known):
    """The same with UT_SPLIT_SCALE_DYNAMIC: every split launch - backbone and regressor - scales by the largest magnitude its
    producer stored in this call."""
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    try:
        eng.set_split_scale("dynamic")
        eng.set_conv_arithmetic("split_f16_always")
        test_model_matches_reference_goldens(eng, golden_dir, known)
        eng.poll_status()
    finally:
        eng.close()


@pytest.mark.parametrize("known", [True, False])
def test_model_matches_reference_goldens(engine, golden_dir, known):
    g = np.load(os.path.join(golden_dir, "model_known.npz" if known else "model_unknown.npz"))
    d = 62 if known else 63
    for si, (feat, pose, raw, mem, ext) in enumerate(_run_steps(engine, known)):
        p = f"s{si}."
        assert np.abs(feat - g[p + "proj"]).max() < 2e-5
        assert np.abs(raw[:, :d] - g[p + "raw"]).max() < 2e-5
        assert np.abs(pose[:, :22] - g[p + "joint_angles"]).max() < ANGLE_TOL
        xf = pose[:, 22:38].reshape(-1, 4, 4)
        assert np.abs(xf[:, :3, :3] - g[p + "wrist_xfs"][:, :3, :3]).max() < 1e-5          # rotation entries
        assert np.abs(xf[:, :3, 3] - g[p + "wrist_xfs"][:, :3, 3]).max() < METRE_TOL
        assert np.array_equal(xf[:, 3], g[p + "wrist_xfs"][:, 3])
        assert np.abs(pose[:, 39:60] - g[p + "sigmas"]).max() < 2e-5
        if not known:
            assert np.abs(pose[:, 38] - g[p + "skel_scales"]).max() < 2e-5
        n = g[p + "mem_state"].shape[0]
        assert mem.shape[0] == n
        assert np.abs(mem - g[p + "mem_state"]).max() < 2e-5
        assert np.abs(ext - g[p + "prev_ext_state"]).max() == 0


def test_unknown_mode_rejects_single_view(engine):
    st = scenarios.model_steps(True)[0]
    feat = engine.backbone(_dev(st["images"]))
    with pytest.raises(AssertionError):
        engine.fuse_temporal_regress(feat, _dev(st["intrinsics"]), _dev(st["extrinsics"]), _dev(st["sample_range"]),
                                     _dev(st["memory_idx"]), _dev(st["use_memory"]), _dev(st["hand_idx"]), 3, False,
                                     None, _native.UT_MODE_UNKNOWN)


def test_keypoints_end_to_end_vs_oracle(engine):
    """network -> pose -> FK keypoints in mm, against the oracle on a larger random batch."""
    s = 24
    rng_imgs = synth.synthetic_crops(2 * s, seed=11)
    k = scenarios._intrinsics("e2e.K", 2 * s, 0)
    x = scenarios._rigid("e2e.X", 2 * s, 0)
    sr = np.array([[2 * i, 2 * i + 2] for i in range(s)], np.int64)
    hand_idx = (np.arange(s) % 2).astype(np.int64)
    axes, rest = scenarios.skeleton_m()
    hm = scenarios.hand_model_mm()
    om = ref_model.OracleModel(synth.synthetic_state_dict(0))
    o = om.forward(torch.from_numpy(rng_imgs), torch.from_numpy(k), torch.from_numpy(x), torch.from_numpy(sr),
                   torch.arange(s), torch.zeros(s, dtype=torch.bool), torch.from_numpy(hand_idx),
                   torch.from_numpy(axes), torch.from_numpy(rest), True)
    xf_mm = o["wrist_xfs"].numpy().copy()
    xf_mm[:, :3, 3] *= 1000.0
    xf_mm[hand_idx == 1, :, 0] *= -1
    want_kp = ref_fk.skin_landmarks(hm, o["joint_angles"].numpy(), xf_mm)
    engine.reset_memory()
    feat = engine.backbone(_dev(rng_imgs))
    pose, _ = engine.fuse_temporal_regress(feat, _dev(k), _dev(x), _dev(sr), torch.arange(s, device=DEV),
                                           torch.zeros(s, dtype=torch.bool, device=DEV), _dev(hand_idx), s, True,
                                           _dev(np.stack([axes, rest])[None]), _native.UT_MODE_KNOWN)
    blob = _dev(_native.hand_model_blob(hm["joint_rotation_axes"], hm["joint_rest_positions"],
                                        hm["landmark_rest_positions"], hm["landmark_rest_bone_weights"],
                                        hm["landmark_rest_bone_indices"])[None])
    kp = engine.fk(blob, pose, pose[:, 22:], mirror=_dev(hand_idx), t_scale=1000.0, ja_stride=60, xf_stride=60, n=s)
    got = pose.cpu().numpy()
    assert np.abs(got[:, :22] - o["joint_angles"].numpy()).max() < ANGLE_TOL
    assert np.abs(kp.cpu().numpy() - want_kp).max() < 1e-3          # mm


def test_fk_matches_stored_reference_keypoints(engine, golden_dir):
    g = np.load(os.path.join(golden_dir, "fk_user05.npz"))
    for rec in ("00", "02", "11"):
        p = f"r{rec}."
        hm = {k[len(p) + 3:]: g[k] for k in g.files if k.startswith(p + "hm.")}
        blob = _dev(_native.hand_model_blob(hm["joint_rotation_axes"], hm["joint_rest_positions"],
                                            hm["landmark_rest_positions"], hm["landmark_rest_bone_weights"],
                                            hm["landmark_rest_bone_indices"])[None])
        ja = g[p + "joint_angles"].astype(np.float32)            # [T,2,22]
        xf = g[p + "wrist_transforms"].astype(np.float32)
        t = ja.shape[0]
        mirror = np.tile(np.array([0, 1], np.int64), t)
        kp = engine.fk(blob, _dev(ja.reshape(-1, 22)), _dev(xf.reshape(-1, 4, 4)), mirror=_dev(mirror))
        kp = kp.cpu().numpy().reshape(t, 2, 21, 3).transpose(1, 0, 2, 3)
        valid = g[p + "valid_tracking"]
        assert np.abs(kp - g[p + "gt_keypoints"])[valid].max() < 1e-3      # mm, vs the reference's stored output
        xfo = xf.copy()
        xfo[:, 1, :, 0] *= -1
        want = ref_fk.skin_landmarks(hm, ja, xfo).transpose(1, 0, 2, 3)
        assert np.abs(kp - want).max() < 2e-4                               # vs the oracle


def test_fk_per_pose_models_and_empty(engine):
    hm = scenarios.hand_model_mm()
    lab = scenarios.labels()
    ja = lab["joint_angles"][:5, 0].astype(np.float32)
    xf = lab["wrist_transforms"][:5, 0].astype(np.float32)
    scales = np.array([0.8, 0.9, 1.0, 1.1, 1.2], np.float32)
    blobs = np.stack([_native.hand_model_blob(hm["joint_rotation_axes"], hm["joint_rest_positions"] * s,
                                              hm["landmark_rest_positions"] * s, hm["landmark_rest_bone_weights"],
                                              hm["landmark_rest_bone_indices"]) for s in scales])
    kp = engine.fk(_dev(blobs), _dev(ja), _dev(xf)).cpu().numpy()
    for i, s in enumerate(scales):
        hmi = dict(hm, joint_rest_positions=hm["joint_rest_positions"] * s,
                   landmark_rest_positions=hm["landmark_rest_positions"] * s)
        assert np.abs(kp[i] - ref_fk.skin_landmarks(hmi, ja[i], xf[i])).max() < 2e-4
    assert engine.fk(_dev(blobs[:1]), torch.empty(0, 22, device=DEV), torch.empty(0, 4, 4, device=DEV)).shape == (0, 21, 3)


def test_fk_multi_bone_blend_and_duplicate_bone_entries(engine, golden_dir):
    """Linear blend skinning with 2-3 non-zero bone weights per landmark (lib/common/hand_skinning.py:56-97).
    In every model the reference ships, landmark 20 (palm centre) blends three bones (weights .887/.077/.036 on
    frames 1, 8, 5) and the other 20 landmarks have one: so the 3-bone blend is pinned by the stored gt_keypoints of
    test_fk_matches_stored_reference_keypoints (checked here for that landmark alone).  This test widens it to every
    landmark on a synthetic model, GPU vs oracle, including two non-zero entries that name the SAME bone: the
    reference's dense scatter `skin_mat[idx] = w` keeps one of them - the last one on the CPU, which is what both
    the oracle and the kernel implement (torch calls duplicate indices in index_put_ undefined: parity unpinned)."""
    g = np.load(os.path.join(golden_dir, "fk_user05.npz"))
    hm0 = {k[len("r00.hm."):]: g[k] for k in g.files if k.startswith("r00.hm.")}
    assert (hm0["landmark_rest_bone_weights"] != 0).sum(1).tolist() == [1] * 20 + [3]
    rng = np.random.default_rng(5)
    n = 64
    w = rng.uniform(0.1, 1.0, (n, 21, 3)).astype(np.float32)
    idx = rng.integers(0, 17, (n, 21, 3)).astype(np.int64)
    w[:, ::3, 2] = 0                      # two bones only
    w[:, 1::6, 0] = 0                     # a zero in front of non-zeros
    idx[:, 2::4, 2] = idx[:, 2::4, 0]     # duplicate bone, both weights non-zero: the later entry wins
    idx[:, 3::7, 1] = idx[:, 3::7, 2]
    w /= w.sum(-1, keepdims=True)
    assert ((w != 0).sum(-1) >= 2).all()
    hm = dict(hm0, landmark_rest_bone_weights=w, landmark_rest_bone_indices=idx,
              joint_rotation_axes=np.broadcast_to(hm0["joint_rotation_axes"], (n, 22, 3)),
              joint_rest_positions=np.broadcast_to(hm0["joint_rest_positions"], (n, 22, 3)),
              landmark_rest_positions=np.broadcast_to(hm0["landmark_rest_positions"], (n, 21, 3)))
    sel = rng.integers(0, g["r00.joint_angles"].shape[0], n)
    ja = g["r00.joint_angles"][sel, 0].astype(np.float32)
    xf = g["r00.wrist_transforms"][sel, 0].astype(np.float32)
    blobs = _native.hand_model_blob(hm["joint_rotation_axes"], hm["joint_rest_positions"], hm["landmark_rest_positions"],
                                    w, idx)
    assert blobs.shape == (n, 321)
    kp = engine.fk(_dev(blobs), _dev(ja), _dev(xf)).cpu().numpy()
    want = ref_fk.skin_landmarks(hm, ja, xf)
    assert np.abs(kp - want).max() < 2e-4                                  # mm
    # the duplicate really matters: dropping the later entry instead changes the result
    w_first = w.copy()
    w_first[:, 2::4, 2] = 0
    other = ref_fk.skin_landmarks(dict(hm, landmark_rest_bone_weights=w_first), ja, xf)
    assert np.abs(other - want)[:, 2::4].max() > 1.0
    # one shared multi-bone model for the whole batch (n_models == 1)
    kp1 = engine.fk(_dev(blobs[:1]), _dev(ja), _dev(xf)).cpu().numpy()
    hm1 = {k: (v[0] if v.ndim == 3 and v.shape[0] == n else v) for k, v in hm.items()}
    assert np.abs(kp1 - ref_fk.skin_landmarks(hm1, ja, xf)).max() < 2e-4
    # the reference's own 3-bone landmark against its stored output
    ja0, xf0 = g["r00.joint_angles"].astype(np.float32), g["r00.wrist_transforms"].astype(np.float32)
    blob0 = _native.hand_model_blob(hm0["joint_rotation_axes"], hm0["joint_rest_positions"], hm0["landmark_rest_positions"],
                                    hm0["landmark_rest_bone_weights"], hm0["landmark_rest_bone_indices"])[None]
    t = ja0.shape[0]
    kp0 = engine.fk(_dev(blob0), _dev(ja0.reshape(-1, 22)), _dev(xf0.reshape(-1, 4, 4)),
                    mirror=_dev(np.tile(np.array([0, 1], np.int64), t))).cpu().numpy().reshape(t, 2, 21, 3)
    valid = g["r00.valid_tracking"].T
    assert np.abs(kp0[:, :, 20] - g["r00.gt_keypoints"].transpose(1, 0, 2, 3)[:, :, 20])[valid].max() < 1e-3


def _rec00_cameras(lab, fi):
    names = ("ImageSizeX", "ImageSizeY", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "k5", "k6")
    return [ref_camera.camera_from_json(dict(zip(names, lab["cameras"][ci])) | {"DistortionModel": "FishEye62"},
                                        lab["camera_to_world_transforms"][fi, ci]) for ci in range(4)]


def test_warp_coordinate_map_equals_reference_goldens(golden_dir):
    """ut_warp_map (the fp32 array the resampler samples with) against the REFERENCE's own maps for 40 crop cameras of
    recording_00 (tests/golden/geometry_rec00.npz: lib/tracker/tracker.py:69-85 through the reference's camera classes),
    given the reference's crop cameras: the same float32 values, entry for entry."""
    from absolutetrack_amd import geometry
    g = dict(np.load(os.path.join(golden_dir, "geometry_rec00.npz")))
    lab = scenarios.labels()
    cams_all, crops_all, src_idx, want = [], [], [], []
    for f, fi in enumerate(g["frames"]):
        cams = _rec00_cameras(lab, int(fi))
        cams_all += [geometry.pack_source_camera(c["f"], c["c"], c["k"], c["T"]) for c in cams]
        for hand in (0, 1):
            for ci in g[f"f{fi}.h{hand}.cams"]:
                ck = f"f{fi}.h{hand}.c{ci}."
                crops_all.append(geometry.pack_crop_camera(tuple(g[ck + "f"]), tuple(g[ck + "c"]), g[ck + "T"]))
                src_idx.append(f * 4 + int(ci))
                want.append(g[ck + "map_sub"])
    got = _native.warp_map(_dev(np.stack(cams_all)), _dev(np.stack(crops_all)), _dev(np.array(src_idx, np.int32)),
                           len(cams_all)).cpu().numpy()[:, ::4, ::4]
    want = np.stack(want)
    assert got.shape == want.shape == (40, 24, 24, 2)
    differ = int((got != want).sum())
    assert differ == 0, (differ, float(np.abs(got - want).max()))


@pytest.mark.parametrize("mode", ["cv2", "float"])
def test_warp_matches_oracle(engine, mode):
    from absolutetrack_amd import geometry
    lab = scenarios.labels()
    hm = scenarios.hand_model_mm()
    frames = synth.synthetic_frames(2, seed=1)                      # [2,4,480,636] u8
    cams_all, crops_all, src_idx, want = [], [], [], []
    for f, fi in enumerate((0, 200)):
        cams = _rec00_cameras(lab, fi)
        for ci, c in enumerate(cams):
            cams_all.append(geometry.pack_source_camera(c["f"], c["c"], c["k"], c["T"]))
        for hand in (0, 1):
            cc = ref_camera.gen_crop_cameras(cams, lab["camera_angles"], hm, lab["joint_angles"][fi, hand],
                                             lab["wrist_transforms"][fi, hand], hand)
            for ci, crop in cc.items():
                crops_all.append(geometry.pack_crop_camera(crop["f"], crop["c"], crop["T"]))
                src_idx.append(f * 4 + ci)
                w = ref_camera.warp_image(cams[ci], crop, frames[f, ci], mode)
                want.append(w.astype(np.float32) / np.float32(255.0))
    got = engine.warp_crops(_dev(frames.reshape(-1, 480, 636)), _dev(np.stack(cams_all)), _dev(np.stack(crops_all)),
                            _dev(np.array(src_idx, np.int32)),
                            _native.UT_REMAP_CV2_FIXED if mode == "cv2" else _native.UT_REMAP_FLOAT).cpu().numpy()
    want = np.stack(want)
    assert got.shape == want.shape == (8, 96, 96)
    diff = np.abs(got - want)
    if mode == "cv2":
        # the same crop cameras on both sides: the coordinate maps are the same float32 values
        # (test_warp_coordinate_map_equals_reference_goldens) and OpenCV's 8-bit remap is integer arithmetic from there
        assert int((diff > 0).sum()) == 0, (int((diff > 0).sum()), float(diff.max()))
    else:
        assert diff.max() < 1e-6, float(diff.max())


@pytest.mark.parametrize("mode", [_native.UT_REMAP_CV2_FIXED, _native.UT_REMAP_FLOAT])
def test_fused_resample_backbone_equals_the_two_calls(engine, mode):
    """ut_warp_backbone (crops kept in the workspace: u8 grey levels in cv2 mode, fp32 in float mode) gives the
    features of ut_warp_crops + ut_backbone bit for bit; a bad src_index is refused the same way."""
    from absolutetrack_amd import geometry
    lab = scenarios.labels()
    hm = scenarios.hand_model_mm()
    frames = synth.synthetic_frames(2, seed=6)
    cams_all, crops_all, src_idx = [], [], []
    for f, fi in enumerate((30, 310)):
        cams = _rec00_cameras(lab, fi)
        cams_all += [geometry.pack_source_camera(c["f"], c["c"], c["k"], c["T"]) for c in cams]
        for hand in (0, 1):
            cc = ref_camera.gen_crop_cameras(cams, lab["camera_angles"], hm, lab["joint_angles"][fi, hand],
                                             lab["wrist_transforms"][fi, hand], hand)
            for ci, crop in cc.items():
                crops_all.append(geometry.pack_crop_camera(crop["f"], crop["c"], crop["T"]))
                src_idx.append(f * 4 + ci)
    src, cam, crop = _dev(frames.reshape(-1, 480, 636)), _dev(np.stack(cams_all)), _dev(np.stack(crops_all))
    idx = _dev(np.array(src_idx, np.int32))
    crops = engine.warp_crops(src, cam, crop, idx, mode)
    want = engine.backbone(crops)
    got = engine.warp_backbone(src, cam, crop, idx, mode)
    assert got.shape == want.shape == (8, 72, 6, 6) and torch.equal(got, want)
    with pytest.raises(IndexError, match="src_index"):
        engine.warp_backbone(src, cam, crop, _dev(np.array([0, 1, 2, 3, 4, 5, 6, 8], np.int32)), mode)


def test_latency_mode_matches_the_default_to_rounding(engine):
    """ut_set_latency_mode: convolutions of a few crops split K across workgroups and add the slabs in a fixed order
    (csrc/ut_api.hip::run_conv).  Same sums in another order: the features agree with the default mode to fp32
    rounding, run to run identically; launches with enough tiles (a batch) are not split and stay bit-identical."""
    fast = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    try:
        fast.set_latency_mode(True)
        for n in (1, 4):
            crops = _dev(synth.synthetic_crops(n, seed=20 + n))
            want = engine.backbone(crops)
            got = fast.backbone(crops)
            scale = float(want.abs().max())
            assert float((got - want).abs().max()) < 2e-5 * max(scale, 1.0), (n, float((got - want).abs().max()), scale)
            assert torch.equal(fast.backbone(crops), got)                 # deterministic
        d = _head_inputs(engine, n_samples=2, seed=5)
        engine.reset_memory()
        fast.reset_memory()
        p0, p1 = _head_call(engine, d, n_slots=2), _head_call(fast, d, n_slots=2)
        assert float((p0 - p1).abs().max()) < 1e-5
        big = _dev(synth.synthetic_crops(300, seed=9))
        assert torch.equal(fast.backbone(big), engine.backbone(big))      # enough tiles: no split
    finally:
        fast.close()


@pytest.mark.parametrize("conv", ["fp32", "split_f16"])
def test_two_backbone_lanes_are_bit_identical(engine, conv):
    """ut_set_backbone_lanes(2): a batch of >= 1024 crops runs as two half-batches on two internal streams (each fills
    the idle tail of the other's launches).  Same kernels on the same crops: same bits, for both crop element types,
    and the caller's stream sees the joined result (no explicit synchronisation here before the comparison).  In split-fp16
    mode too: the calibrated activation scales do not depend on what a launch holds.  With dynamic scales each half-batch takes
    its own, and the lanes agree with one lane to the split arithmetic's rounding only."""
    one = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    two = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    try:
        one.set_conv_arithmetic(conv)
        two.set_conv_arithmetic(conv)
        two.set_backbone_lanes(2)
        g = torch.Generator(device=DEV)
        g.manual_seed(31)
        for n in (1024, 1501):
            crops = torch.randint(0, 256, (n, 96, 96), device=DEV, generator=g, dtype=torch.uint8).float() / 255.0
            want = one.backbone(crops)
            got = two.backbone(crops)
            assert torch.equal(got, want), n
            if conv == "fp32":
                assert torch.equal(want, engine.backbone(crops))
        assert torch.equal(two.backbone(crops[:700]), want[:700])      # below the lane threshold: one lane
        two.poll_status()
        if conv != "fp32":
            one.set_split_scale("dynamic")
            two.set_split_scale("dynamic")
            a, b = one.backbone(crops), two.backbone(crops)
            assert (a - b).abs().max().item() < 2e-6 * max(1.0, a.abs().max().item())
            assert (a - want).abs().max().item() < 2e-6 * max(1.0, a.abs().max().item())
            two.poll_status()
    finally:
        one.close()
        two.close()


def test_split_f16_calibrated_scales_batch_independence_and_range_guard(engine):
    """Calibrated activation scales (the default of the split-fp16 mode; include/umetrack_hip.h::ut_set_split_scale):
    (i) a crop's features do not depend on its batch - alone, in any sub-batch, at any pass size, next to much brighter crops:
    the same bits (with dynamic scales the smallest activations of a dark crop round differently next to bright ones);  (ii) two handles calibrate to the same words (the built-in set is generated on the
    device, identically everywhere);  (iii) inputs far beyond the calibrated range are reported, not silently saturated: after a
    calibration on crops of 1/64 the brightness, crops 64 x brighter still (4096 x the calibration maximum) raise "range check"
    at the next status read - and run clean with dynamic scales and after a calibration that covers them."""
    crops = _dev(synth.synthetic_crops(40, seed=17))
    eng = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    other = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    try:
        eng.set_conv_arithmetic("split_f16_always")
        other.set_conv_arithmetic("split_f16_always")
        cal = eng.split_calibration()
        assert np.array_equal(cal, other.split_calibration()) and np.isfinite(cal).all()
        assert (cal[:24] > 0).all() and cal[24] == 0 and (cal[25:] > 0).all()      # (tensor 24 feeds the projection: no split consumer)
        whole = eng.backbone(crops)
        assert torch.equal(eng.backbone(crops[7:8]), whole[7:8])
        assert torch.equal(eng.backbone(crops[11:29]), whole[11:29])
        eng.set_backbone_chunk(9)
        assert torch.equal(eng.backbone(crops), whole)
        eng.set_backbone_chunk(0)
        fp32 = engine.backbone(crops)
        scale = max(1.0, fp32.abs().max().item())
        assert (whole - fp32).abs().max().item() < 1e-5 * scale
        eng.poll_status()
        eng.set_split_scale("dynamic")
        dyn = eng.backbone(crops)
        assert (dyn - whole).abs().max().item() < 2e-6 * scale
        eng.set_split_scale("calibrated")
        dim = torch.cat([crops[:1] * 0.02, crops[1:]])                      # one dark crop among normal ones
        assert torch.equal(eng.backbone(dim[:1]), eng.backbone(dim)[:1])
        # (iii)
        eng.calibrate_split(crops / 64.0)
        assert not np.array_equal(eng.split_calibration(), cal)
        eng.backbone(crops / 64.0)
        eng.poll_status()
        eng.backbone(crops * 64.0)
        with pytest.raises(FloatingPointError, match="calibrated range"):
            eng.poll_status()
        eng.set_split_scale("dynamic")
        want = engine.backbone(crops * 64.0)
        got = eng.backbone(crops * 64.0)
        eng.poll_status()
        assert (got - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
        eng.set_split_scale("calibrated")
        eng.calibrate_split(crops * 64.0)
        got = eng.backbone(crops * 64.0)
        eng.poll_status()
        assert (got - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
        eng.calibrate_split()                                               # back to the built-in set
        assert np.array_equal(eng.split_calibration(), cal)
        assert torch.equal(eng.backbone(crops), whole)
    finally:
        eng.close()
        other.close()


def _head_inputs(engine, n_samples=3, seed=2):
    n = 2 * n_samples
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    feat = engine.backbone(_dev(synth.synthetic_crops(n, seed=seed)))
    k = torch.eye(3, device=DEV).repeat(n, 1, 1)
    k[:, 0, 0] = k[:, 1, 1] = 100 + 60 * torch.rand(n, device=DEV, generator=g)
    k[:, 0, 2] = k[:, 1, 2] = 47.5
    x = torch.eye(4, device=DEV).repeat(n, 1, 1)
    x[:, :3, 3] = torch.rand(n, 3, device=DEV, generator=g) * 0.2
    sr = torch.arange(0, n, 2, device=DEV)[:, None] + torch.tensor([0, 2], device=DEV)
    hm = scenarios.hand_model_mm()
    skel = _dev(np.stack([hm["joint_rotation_axes"], hm["joint_rest_positions"] * np.float32(0.001)])[None].astype(np.float32))
    return dict(feat=feat, k=k, x=x, sr=sr, mem=torch.arange(n_samples, device=DEV), use=torch.zeros(n_samples, dtype=torch.bool, device=DEV),
                hand=(torch.arange(n_samples, device=DEV) % 2), skel=skel, s=n_samples)


def _head_call(engine, d, mode=_native.UT_MODE_KNOWN, **over):
    a = dict(d, **over)
    return engine.fuse_temporal_regress(a["feat"], a["k"], a["x"], a["sr"], a["mem"], a["use"], a["hand"],
                                        over.get("n_slots", a["s"]), over.get("all_multiview", True),
                                        a["skel"] if mode == _native.UT_MODE_KNOWN else None, mode)[0].clone()


def test_index_checks_reject_bad_descriptors(engine):
    """What the reference raises IndexError / AssertionError on in Python (lib/tracker/tracker.py:330,
    lib/models/temporal.py:101-137, lib/models/umetrack_model.py:149-166,224-229) comes back as UT_E_INVALID /
    UT_E_UNSUPPORTED from the device-side checks: nothing is read or written out of range and the temporal state is
    the same afterwards."""
    engine.reset_memory()
    d = _head_inputs(engine)
    s = d["s"]
    good = _head_call(engine, d)
    mem0, ext0 = [t.clone() for t in engine.get_memory()]
    t = lambda v: torch.tensor(v, device=DEV)
    bad_cases = {
        "range width 3": dict(sr=t([[0, 3], [2, 4], [4, 6]])),
        "range width 0": dict(sr=t([[0, 2], [3, 3], [4, 6]])),
        "range negative": dict(sr=t([[-1, 1], [2, 4], [4, 6]])),
        "range past n_crops": dict(sr=t([[0, 2], [2, 4], [5, 7]])),
        "slot == n_slots": dict(mem=t([0, 1, 3])),
        "slot negative": dict(mem=t([0, -1, 2])),
        "slot far out": dict(mem=t([0, 1, 1 << 40])),
        "duplicate slot": dict(mem=t([0, 1, 1])),
        "hand 2": dict(hand=t([0, 1, 2])),
        "hand -1": dict(hand=t([-1, 1, 0])),
    }
    for name, over in bad_cases.items():
        with pytest.raises(IndexError, match="index check"):
            _head_call(engine, d, **over)
        mem1, ext1 = engine.get_memory()
        assert torch.equal(mem1, mem0) and torch.equal(ext1, ext0), name       # state untouched
    # unknown-skeleton mode: a one-view sample hidden behind n_crops == 2 * n_samples is found per sample
    with pytest.raises(AssertionError, match="single-view"):
        _head_call(engine, d, _native.UT_MODE_UNKNOWN, sr=t([[0, 1], [1, 3], [4, 6]]))
    with pytest.raises(AssertionError, match="single-view"):
        _head_call(engine, d, _native.UT_MODE_UNKNOWN, all_multiview=False)
    # ... and is fine with the known skeleton (the single-view branch)
    assert torch.isfinite(_head_call(engine, d, sr=t([[0, 1], [1, 3], [4, 6]]))).all()
    engine.reset_memory()
    assert torch.equal(_head_call(engine, d), good)
    # resampler: src_index outside the source stack
    src = torch.zeros(2, 480, 636, dtype=torch.uint8, device=DEV)
    cam = torch.zeros(2, 32, dtype=torch.float64, device=DEV)
    cam[:, 0:4] = torch.tensor([240.0, 240.0, 317.5, 239.5], dtype=torch.float64)
    cam[:, 12] = cam[:, 16] = cam[:, 20] = 1
    crop = torch.zeros(3, 24, dtype=torch.float64, device=DEV)
    crop[:, 0:4] = torch.tensor([240.0, 240.0, 47.5, 47.5], dtype=torch.float64)
    crop[:, 4] = crop[:, 8] = crop[:, 12] = 1
    for idx in ([0, 2, 1], [0, -1, 1], [1 << 30, 0, 0]):
        with pytest.raises(IndexError, match="src_index"):
            engine.warp_crops(src, cam, crop, torch.tensor(idx, dtype=torch.int32, device=DEV))
    assert engine.warp_crops(src, cam, crop, torch.tensor([0, 1, 1], dtype=torch.int32, device=DEV)).shape == (3, 96, 96)
    # stateless entry (no handle): always synchronous
    with pytest.raises(RuntimeError, match="src_index"):
        rc = engine.lib.ut_warp_crops(None, _native._ptr(src), 2, 480, 636, _native._ptr(cam), _native._ptr(crop),
                                      _native._ptr(torch.tensor([0, 5, 1], dtype=torch.int32, device=DEV)), 3, 0,
                                      _native._ptr(torch.empty(3, 96, 96, device=DEV)), _native._stream(torch.device(DEV)))
        if rc:
            raise RuntimeError(engine.lib.ut_last_error(None).decode())


def test_deferred_index_checks(engine):
    """UT_CHECK_DEFERRED: the call itself does not synchronise or raise; the bad call's work is skipped on the device
    (state and the rest of the batch untouched), ut_poll_status reports it once, later calls run normally."""
    engine.reset_memory()
    d = _head_inputs(engine)
    good = _head_call(engine, d)
    mem0, ext0 = [t.clone() for t in engine.get_memory()]
    engine.set_index_checks(deferred=True)
    try:
        engine.poll_status()                                  # clean
        snap = torch.zeros(2, dtype=torch.int32, device=DEV)
        engine.status_snapshot(snap)
        assert snap.tolist() == [0, 0]
        _head_call(engine, d, mem=torch.tensor([0, 1, 7], device=DEV))
        _head_call(engine, d)                                  # still flagged: skipped as well
        engine.status_snapshot(snap)                           # ut_status_snapshot: the verdict as a stream-ordered device copy
        assert snap[0].item() != 0
        mem1, ext1 = engine.get_memory()
        assert torch.equal(mem1, mem0) and torch.equal(ext1, ext0)
        with pytest.raises(IndexError, match="memory_idx"):
            engine.poll_status()
        engine.poll_status()                                  # reported once
        engine.reset_memory()
        assert torch.equal(_head_call(engine, d), good)
        engine.poll_status()
    finally:
        engine.set_index_checks(deferred=False)


def test_two_handles_on_one_device_are_independent(engine):
    """A second handle in the same process (the dynamic-LDS attribute of the conv kernels is kept per device, temporal
    state and workspace per handle); calls with another current stream / interleaved order give identical results."""
    other = _native.HipEngine(synth.synthetic_state_dict(0), DEV)
    try:
        crops = _dev(synth.synthetic_crops(6, seed=12))
        a = engine.backbone(crops)
        b = other.backbone(crops)
        assert torch.equal(a, b)
        d = _head_inputs(engine)
        engine.reset_memory()
        other.reset_memory()
        p1 = _head_call(engine, d)
        q1 = _head_call(other, d)
        p2 = _head_call(engine, d, use=torch.ones(3, dtype=torch.bool, device=DEV))     # uses engine's own memory
        assert torch.equal(p1, q1) and not torch.equal(p2, p1)
        assert other.get_memory()[0].shape[0] == 3
        q2 = _head_call(other, d, use=torch.ones(3, dtype=torch.bool, device=DEV))
        assert torch.equal(p2, q2)
    finally:
        other.close()
    assert torch.equal(engine.backbone(crops), a)           # the first handle outlives the second


def test_conv_tile_shapes_agree(engine):
    """The convolution dispatch picks its tile shape from the launch size alone (csrc/conv_igemm.hip::launch_conv_igemm):
    the head's 3x3 / 1x1 convolutions run on 64x128 tiles up to 3 x 256 full-height tiles and on 128x128 tiles beyond
    (S > 2730 samples), the backbone on 128x64 / 128x128 / the halo-patch kernel by channel count.  Every shape walks K
    in the same order with the same MFMA, so a sample's result must not depend on which shape its batch selected: bit
    for bit, big batch vs the same samples in small batches."""
    s_big = 4800
    n = 2 * s_big
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    feat = torch.randn(n, 72, 6, 6, device=DEV, generator=g) * 0.3
    k = torch.eye(3, device=DEV).repeat(n, 1, 1)
    k[:, 0, 0] = k[:, 1, 1] = 100 + 60 * torch.rand(n, device=DEV, generator=g)
    k[:, 0, 2] = k[:, 1, 2] = 47.5
    x = torch.eye(4, device=DEV).repeat(n, 1, 1)
    x[:, :3, 3] = torch.rand(n, 3, device=DEV, generator=g) * 0.2
    hm = scenarios.hand_model_mm()
    skel = _dev(np.stack([hm["joint_rotation_axes"], hm["joint_rest_positions"] * np.float32(0.001)])[None].astype(np.float32))

    def run(lo, hi):
        s = hi - lo
        sr = torch.arange(0, 2 * s, 2, device=DEV)[:, None] + torch.tensor([0, 2], device=DEV)
        engine.reset_memory()
        return engine.fuse_temporal_regress(feat[2 * lo:2 * hi], k[2 * lo:2 * hi], x[2 * lo:2 * hi], sr,
                                            torch.arange(s, device=DEV), torch.zeros(s, dtype=torch.bool, device=DEV),
                                            (torch.arange(lo, hi, device=DEV) % 2), s, True, skel, _native.UT_MODE_KNOWN)[0].clone()
    big = run(0, s_big)
    assert torch.isfinite(big).all()
    for lo, hi in ((0, 7), (1000, 1512), (s_big - 300, s_big)):
        assert torch.equal(run(lo, hi), big[lo:hi]), (lo, hi)
    engine.reset_memory()
