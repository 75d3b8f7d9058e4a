"""Latency of the drop-in per-frame API (the loop body of run_eval_known_skeleton.py:68-89 through lib.*):
gen_crop_cameras -> track_frame -> landmarks_from_hand_pose for both hands, one frame at a time, images handed
over as host numpy arrays (so this is the PCIe-inclusive, launch-latency-bound figure; the batched path is bench.py).
    python tools/bench_tracker.py [n_frames] [--profile]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from absolutetrack_amd import pipeline, synth  # noqa: E402


def main():
    import faulthandler
    faulthandler.dump_traceback_later(300, exit=True)      # a hung launch sequence shows where, then exits
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
    from lib.models.umetrack_model import UmeTrackModel
    from lib.tracker.perspective_crop import landmarks_from_hand_pose
    from lib.tracker.tracker import HandTracker, HandTrackerOpts, InputFrame, ViewData
    from lib.tracker.tracking_result import SingleHandPose
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    model = UmeTrackModel(synth.synthetic_state_dict(0))
    model.eval()
    trk = HandTracker(model, HandTrackerOpts())
    frames = synth.synthetic_frames(8, seed=2)
    angles = list(lab["camera_angles"])

    def one(fi):
        cams = pipeline.cameras_for_frame(lab, fi % 369)
        sample = InputFrame(views=[ViewData(image=frames[fi % 8, ci], camera=cams[ci], camera_angle=angles[ci]) for ci in range(4)])
        gt = {h: SingleHandPose(joint_angles=lab["joint_angles"][fi % 369, h], wrist_xform=lab["wrist_transforms"][fi % 369, h],
                                hand_confidence=1.0) for h in (0, 1)}
        cc = trk.gen_crop_cameras(cams, angles, hm, gt, min_num_crops=1)
        res = trk.track_frame(sample, hm, cc)
        out = {}
        for h in res.hand_poses:
            out[h] = (landmarks_from_hand_pose(hm, res.hand_poses[h], h), landmarks_from_hand_pose(hm, gt[h], h))
        return out

    for fi in range(10):
        one(fi)
    torch.cuda.synchronize()
    if "--profile" in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for fi in range(10, 10 + n):
            one(fi)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
        return
    t0 = time.perf_counter()
    for fi in range(10, 10 + n):
        one(fi)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"per-frame API: {dt / n * 1e3:.2f} ms per 2-hand frame = {n / dt:.0f} frames/s = {2 * n / dt:.0f} hand-frames/s "
          f"(reference on CPU: ~40-80 ms per frame, SURVEY 8 d)")


if __name__ == "__main__":
    main()
