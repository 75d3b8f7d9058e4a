"""GPU parity of ut_keypoint_metrics against the reference's load_eval._compute_metrics
outputs (tests/golden/metrics.npz, from oracle/gen_goldens.py).  float64 arithmetic on float32-valued inputs in the
reference's order: tolerance 1e-12 relative (sqrt / summation-order ulps)."""
import os

import numpy as np
import pytest
import torch

from absolutetrack_amd import _native, metrics
from oracle import scenarios

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_compute_metrics_matches_reference(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "metrics.npz")))
    c = scenarios.metrics_case()
    m = metrics._compute_metrics(c["gt_keypoints"], c["tracked_keypoints"], c["valid_tracking"])
    assert m.keypoint_errors.dtype == np.float64 and m.keypoint_errors.shape == g["keypoint_errors"].shape
    np.testing.assert_allclose(m.keypoint_errors, g["keypoint_errors"], rtol=1e-12)
    np.testing.assert_allclose(m.keypoint_accelerations, g["keypoint_accelerations"], rtol=1e-12)
    np.testing.assert_allclose(m.gt_keypoint_accelerations, g["gt_keypoint_accelerations"], rtol=1e-12)
    pck = metrics.PCK_curve(m.keypoint_errors, metrics.PCK_THRESHOLDS) * 100.0
    np.testing.assert_allclose(pck, g["pck"], atol=1e-9)
    assert abs(float(metrics.normalized_AUC(metrics.PCK_THRESHOLDS, pck)) - float(g["auc"])) < 1e-12


def test_aggregate_over_result_files(tmp_path, golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "metrics.npz")))
    c = scenarios.metrics_case()
    for name in ("a/recording_00.npy", "b/recording_01.npy"):
        metrics.save_eval_results(str(tmp_path / name), c["tracked_keypoints"], c["gt_keypoints"], c["valid_tracking"])
    out = metrics.aggregate_metrics(str(tmp_path), verbose=False)
    assert out["n_total"] == 2 * c["valid_tracking"].size and out["n_valid"] == 2 * int(c["valid_tracking"].sum())
    assert abs(out["mean_keypoint_error"] - g["keypoint_errors"].mean()) < 1e-9
    assert abs(out["auc_score"] - float(g["auc"])) < 1e-9      # two copies of the same errors: same curve
    assert abs(out["mean_keypoint_acceleration"] - g["keypoint_accelerations"].mean()) < 1e-9
    assert metrics.aggregate_metrics(str(tmp_path / "nothing_here"), verbose=False) is None


def test_short_sequences_and_argument_checks():
    gt = torch.rand(2, 2, 21, 3, device=DEV)
    m = _native.keypoint_metrics(gt, gt + 1.0, torch.ones(2, 2, dtype=torch.bool, device=DEV))
    assert m["acc"].shape == (2, 0) and torch.allclose(m["err"], torch.full((2, 2), 3.0 ** 0.5, dtype=torch.float64, device=DEV))
    with pytest.raises(ValueError):
        _native.keypoint_metrics(gt, gt[:, :1], torch.ones(2, 2, dtype=torch.bool, device=DEV))
    with pytest.raises(ValueError):
        _native.keypoint_metrics(gt, gt, torch.ones(2, 3, dtype=torch.bool, device=DEV))
    with pytest.raises(_native.NativeLibraryError):
        _native.keypoint_metrics(gt.cpu(), gt.cpu(), torch.ones(2, 2, dtype=torch.bool))
    lib = _native.load_library()
    assert lib.ut_keypoint_metrics(None, None, None, None, 1, 5, None, None, None, None, None) != 0
    assert b"ut_keypoint_metrics" in lib.ut_last_error(None)
