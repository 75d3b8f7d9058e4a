"""CPU tests of the host-side rows f3 (on-disk formats) and f4 (metric helpers, result files) - no GPU needed.
Fixtures: tests/golden/*.torch.{idx,bin} were written by the product's writer and verified readable by the
reference's own lib.data_utils.idxbinfile.TorchIdx in oracle/gen_goldens.py; tests/golden/metrics.npz holds the
outputs of the reference's load_eval._compute_metrics / lib.common.metric_utils on scenarios.metrics_case()."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from absolutetrack_amd import formats, metrics, pipeline
from oracle import scenarios


# ----------------------------------------------------------------------------- f3: .torch.idx / .torch.bin
def test_idxbin_fixtures_parse_to_the_scenario(golden_dir):
    c = scenarios.idxbin_case()
    mono = formats.TorchIdx(os.path.join(golden_dir, "seq_mono.torch.idx"))
    assert mono.is_uniform and mono.shape == (3, 2, 2, 16, 24) and mono.dtype == np.uint8 and len(mono) == 3
    assert mono.item_shape() == (2, 2, 16, 24) and mono.data_size_bytes() == c["mono"].nbytes
    assert np.array_equal(mono.read_bin(), c["mono"])
    lab = formats.TorchIdx(os.path.join(golden_dir, "seq_labels.torch.idx"))
    assert not lab.is_uniform and lab.shape is None and lab.dtype == np.dtype("object")
    assert lab.read_bin() == c["labels"]
    rag = formats.TorchIdx(os.path.join(golden_dir, "ragged_f32.torch.idx"))
    got = rag.read_bin()
    assert rag.shape is None and [g.shape for g in got] == [(2, 3), (4,), (2, 3, 4)]
    assert all(np.array_equal(g, w) and g.dtype == np.float32 for g, w in zip(got, c["ragged"]))
    with pytest.raises(ValueError):
        rag.item_shape()
    assert rag.item_shape(2) == (2, 3, 4)
    seq = formats.read_sequence(os.path.join(golden_dir, "seq_mono.torch.idx"),
                                os.path.join(golden_dir, "seq_labels.torch.idx"), 1)
    assert np.array_equal(seq["mono"], c["mono"][1]) and seq["labels"] == c["labels"][1]


def test_idxbin_byte_offsets_match_reference(golden_dir):
    """byte_offset / byte_offsets against the reference's own TorchIdx on the three fixtures (tests/golden/idxbin_offsets.npz,
    written by oracle/gen_goldens.py from lib/data_utils/idxbinfile.py:196-231): `end == -1` stops before the end-of-data
    offset on the uniform and on the ragged branch, `end == N + 1` includes it."""
    g = np.load(os.path.join(golden_dir, "idxbin_offsets.npz"))
    for name in ("seq_mono", "seq_labels", "ragged_f32"):
        idx = formats.TorchIdx(os.path.join(golden_dir, name + ".torch.idx"))
        n = int(g[name + ".n"])
        assert len(idx) == n
        assert np.array_equal(np.asarray(idx.byte_offsets(0, -1), np.int64), g[name + ".to_minus1"])
        assert np.array_equal(np.asarray(idx.byte_offsets(0, n + 1), np.int64), g[name + ".to_n_plus_1"])
        assert np.array_equal(np.asarray(idx.byte_offsets(1, -1), np.int64), g[name + ".from1_minus1"])
        assert np.array_equal(np.asarray(idx.byte_offsets(1, n), np.int64), g[name + ".from1_to_n"])
        assert [idx.byte_offset(i) for i in list(range(n + 1)) + [-1]] == g[name + ".single"].tolist()


def test_idxbin_round_trip_and_errors(tmp_path):
    rng = np.random.default_rng(0)
    for dt in ("uint8", "int8", "int16", "int32", "int64", "float32", "float64"):
        a = (rng.random((4, 3, 5)) * 100).astype(dt)
        p = str(tmp_path / f"a_{dt}.torch.idx")
        formats.write_torch_idx_bin(p, a)
        idx = formats.TorchIdx(p)
        assert idx.shape == (4, 3, 5) and idx.dtype == np.dtype(dt) and np.array_equal(idx.read_bin(), a)
        raw = open(idx.bin_path, "rb").read()
        assert np.array_equal(idx.view_buffer_at(2, raw), a[2])
    p = str(tmp_path / "a_float32.torch.idx")
    good = np.frombuffer(open(p, "rb").read(), np.int64).copy()
    bad = good.copy()
    bad[0] = 1234
    with pytest.raises(ValueError, match="bad magic"):
        formats.TorchIdx(p, buffer=bad.tobytes())
    bad = good.copy()
    bad[1] = 7
    with pytest.raises(ValueError, match="unsupported version"):
        formats.TorchIdx(p, buffer=bad.tobytes())
    bad = good.copy()
    bad[2] = 42
    with pytest.raises(KeyError):
        formats.TorchIdx(p, buffer=bad.tobytes())
    bad = good.copy()
    bad[3] = 8
    with pytest.raises(ValueError, match="item size"):
        formats.TorchIdx(p, buffer=bad.tobytes())
    with pytest.raises(ValueError, match="invalid length"):
        formats.TorchIdx(p, buffer=good.tobytes()[:-3])
    with pytest.raises(ValueError, match="expected"):
        formats.TorchIdx(p).view_buffer(b"\0" * 5)
    with pytest.raises(ValueError):
        formats.write_torch_idx_bin(str(tmp_path / "x.torch.idx"), [np.zeros(2, np.float32), np.zeros(2, np.int32)])
    with pytest.raises(ValueError):
        formats.write_torch_idx_bin(str(tmp_path / "y.torch.idx"), [np.zeros(2, np.float16)])
    formats.write_torch_idx_bin(str(tmp_path / "e.torch.idx"), [])
    assert len(formats.TorchIdx(str(tmp_path / "e.torch.idx"))) == 0


# ----------------------------------------------------------------------------- f3: label JSON, frame stream
def _label_json(lab, n_frames=5):
    cams = [dict(zip(pipeline._CAM_FIELDS, (float(v) for v in lab["cameras"][ci])), DistortionModel="FishEye62")
            for ci in range(4)]
    for c in cams:
        c["ImageSizeX"], c["ImageSizeY"] = int(c["ImageSizeX"]), int(c["ImageSizeY"])
    hm = {k[3:]: v.tolist() for k, v in lab.items() if k.startswith("hm.")}
    hm.update(joint_frame_index=[0] * 22, joint_parent=[0] * 22, joint_first_child=[0] * 22,
              joint_next_sibling=[0] * 22, hand_scale=None)
    conf = lab["hand_confidences"][:n_frames].copy()
    conf[1, 1] = 0.0
    return {"cameras": cams, "camera_angles": lab["camera_angles"].tolist(), "hand_model": hm,
            "joint_angles": lab["joint_angles"][:n_frames].tolist(),
            "wrist_transforms": lab["wrist_transforms"][:n_frames].tolist(), "hand_confidences": conf.tolist(),
            "camera_to_world_transforms": lab["camera_to_world_transforms"][:n_frames].tolist()}


def test_label_json_and_synced_stream(tmp_path):
    lab = pipeline.load_labels()
    js = _label_json(lab)
    p = tmp_path / "rec.json"
    p.write_text(json.dumps(js))
    hp = formats._load_hand_pose_labels(str(p))
    assert len(hp) == 5 and len(hp.cameras) == 4 and hp.cameras[0].width == 636 and hp.cameras[0].height == 480
    assert hp.hand_model.joint_rotation_axes.dtype == torch.float32 and hp.hand_model.hand_scale is None
    assert hp.joint_angles.shape == (5, 2, 22) and hp.joint_angles.dtype == np.float64
    arr = formats.labels_to_arrays(hp)
    for k in ("cameras", "camera_angles"):
        assert np.array_equal(arr[k], lab[k]), k
    for k in ("joint_angles", "wrist_transforms", "camera_to_world_transforms"):
        assert np.array_equal(arr[k], lab[k][:5]), k
    assert np.array_equal(arr["hm.joint_rest_positions"], lab["hm.joint_rest_positions"])
    # stream over decoded frames: [H, 4*W] mono -> 4 views, gt only for confident hands
    frames = [np.arange(480 * 4 * 636, dtype=np.uint32).reshape(480, 4 * 636).astype(np.uint8) + i for i in range(5)]
    stream = formats.SyncedImagePoseStream(str(tmp_path / "rec.mp4"), frames=frames)
    assert len(stream) == 5
    items = list(stream)
    frame1, gt1 = items[1]
    assert sorted(gt1) == [0] and sorted(items[0][1]) == [0, 1]
    assert len(frame1.views) == 4 and frame1.views[2].image.shape == (480, 636)
    assert np.array_equal(frame1.views[2].image, frames[1][:, 2 * 636:3 * 636])
    assert np.array_equal(frame1.views[2].camera.camera_to_world_xf, lab["camera_to_world_transforms"][1, 2])
    assert frame1.views[3].camera_angle == lab["camera_angles"][3]
    assert np.array_equal(gt1[0].joint_angles, lab["joint_angles"][1, 0])
    with pytest.raises(AssertionError):
        formats.SyncedImagePoseStream(str(tmp_path / "rec.mp4"), frames=frames[:3])
    with pytest.raises(ImportError):      # PyAV is not in this image; the module itself must still import
        len(formats.VideoStream("x.mp4"))
    import lib.tracker.video_pose_data as vpd
    import lib.data_utils.idxbinfile as ib
    assert vpd.SyncedImagePoseStream is formats.SyncedImagePoseStream and ib.TorchIdx is formats.TorchIdx


# ----------------------------------------------------------------------------- f4: PCK / AUC / result files
def test_pck_and_auc_match_reference_goldens(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "metrics.npz")))
    c = scenarios.metrics_case()
    pck = metrics.PCK_curve(g["keypoint_errors"], metrics.PCK_THRESHOLDS) * 100.0
    assert np.array_equal(pck, g["pck"])
    assert float(metrics.normalized_AUC(metrics.PCK_THRESHOLDS, pck)) == float(g["auc"])
    err = np.linalg.norm(c["gt_keypoints"] - c["tracked_keypoints"], axis=-1)
    mask = np.repeat(c["valid_tracking"][..., None], 21, -1).astype(np.float64)
    per_hand = metrics.PCK_curve(err, metrics.PCK_THRESHOLDS, mask=mask, axis=0)
    assert np.array_equal(per_hand, g["pck_per_hand"])
    assert np.array_equal(metrics.normalized_AUC(metrics.PCK_THRESHOLDS, per_hand), g["auc_per_hand"])
    assert 0 < pck[20] < pck[-1] <= 100          # the curve is neither empty nor saturated
    # an all-invalid row: division guarded like the reference's _safe_div
    z = metrics.PCK_curve(err[:1], metrics.PCK_THRESHOLDS, mask=np.zeros_like(err[:1]), axis=0)
    assert (z == 0).all()
    import lib.common.metric_utils as mu
    assert mu.PCK_curve is metrics.PCK_curve and mu.MAX_LANDMARK_ERROR_MM == 50


def test_result_file_format_round_trip(tmp_path):
    c = scenarios.metrics_case()
    p = str(tmp_path / "out" / "user05" / "recording_00.npy")
    metrics.save_eval_results(p, c["tracked_keypoints"], c["gt_keypoints"], c["valid_tracking"])
    with open(p, "rb") as f:            # what the reference's load_eval.py does with the file (our own file)
        raw = pickle.load(f)
    assert sorted(raw) == ["gt_keypoints", "tracked_keypoints", "valid_tracking"]
    back = metrics.load_eval_results(p)
    for k in raw:
        assert np.array_equal(back[k], c[k]) and back[k].dtype == c[k].dtype
    evil = str(tmp_path / "evil.npy")
    with open(evil, "wb") as f:
        pickle.dump({"tracked_keypoints": os.path.join, "gt_keypoints": 1, "valid_tracking": 2}, f)
    with pytest.raises(pickle.UnpicklingError):
        metrics.load_eval_results(evil)
    other = str(tmp_path / "other.npy")
    with open(other, "wb") as f:
        pickle.dump({"a": np.zeros(2)}, f)
    with pytest.raises(ValueError):
        metrics.load_eval_results(other)
    if not torch.cuda.is_available():
        from absolutetrack_amd import _native
        with pytest.raises(_native.NativeLibraryError):
            metrics.aggregate_metrics(str(tmp_path / "out"), verbose=False)
