"""recording_00 end to end, as a sequence: the `north_star` accuracy target ("MPJPE on sample_data/recording_00 within
0.05 mm of the reference") over all 369 label frames x 2 hands, driven through the drop-in `lib.tracker.tracker.HandTracker`
exactly like the reference's eval scripts drive it, with the temporal memory engaged from the second frame on
(`memory_idx = hand_idx`, `use_memory` from the validity history).

  known skeleton    run_eval_known_skeleton.py:68-93
  unknown skeleton  run_eval_unknown_skeleton.py:49-78 (calibrate the generic skeleton's scale on the first 30 hand samples),
                    :99-126 (reset the history and re-track with the calibrated skeleton)

The reference's pixels and weights are missing blobs, so the reference side is the CPU oracle (pinned to
the reference by tests/test_oracle_pinning.py) fed the product's crops of seeded synthetic images, with the seeded
synthetic weights; each side keeps its own temporal state, validity history and calibrated scale."""
import pytest
import torch

from absolutetrack_amd import synth
from oracle import checks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def weights():
    torch.set_num_threads(min(torch.get_num_threads(), 16))   # the oracle's 4-crop forwards crawl when oversubscribed
    return synth.synthetic_state_dict(0)


def _check(r):
    assert r["frames"] == 369 and r["hand_frames"] == 738, r
    assert r["mpjpe_delta_mm"] < 0.05, r                       # BASELINE.json north_star
    assert r["max_joint_angle_err_rad"] < 1e-4, r
    assert r["max_wrist_translation_err_mm"] < 1e-3, r
    assert r["max_keypoint_err_mm"] < 1e-3, r


def test_recording00_known_skeleton_sequence(weights):
    r = checks.run_recording00(weights, "cuda:0", known=True)
    print("recording_00 known:", r)
    _check(r)


def test_recording00_unknown_skeleton_two_pass(weights):
    r = checks.run_recording00(weights, "cuda:0", known=False)
    print("recording_00 unknown:", r)
    assert r["calibration_samples"] == 30
    assert r["scale_mean_abs_diff"] < 2e-5 and r["scale_max_abs_diff"] < 2e-5, r
    _check(r)
