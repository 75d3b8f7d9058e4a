"""Time the batched crop-camera generator (row f1) against the per-frame host path it replaces.
usage: python tools/bench_cropgen.py [n_frames]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from absolutetrack_amd import pipeline  # noqa: E402


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    frames = list(range(n_frames))
    pipeline.crop_plan_on_device(lab, hm, frames[:8], "cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan = pipeline.crop_plan_on_device(lab, hm, frames, "cuda:0")
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    # kernel alone
    from absolutetrack_amd import _native
    c = pipeline.label_candidates(lab, frames)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    blob = torch.from_numpy(_native.hand_model_blob(hm.joint_rotation_axes, hm.joint_rest_positions, hm.landmark_rest_positions,
                                                    hm.landmark_rest_bone_weights, hm.landmark_rest_bone_indices)).reshape(1, 321).to("cuda:0")
    a = (t(c["cam_params"]), t(c["camera_angles"]), blob, hm.joint_limits.float().to("cuda:0"), t(c["joint_angles"]),
         t(c["wrist_xf"]), t(c["frame_idx"]), t(c["hand_idx"]), c["n_cams"], c["src_wh"])
    _native.gen_crop_cameras(*a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _native.gen_crop_cameras(*a)
    e1.record()
    torch.cuda.synchronize()
    t_k = e0.elapsed_time(e1) / 10
    n_host = min(n_frames, 64)
    pipeline.crop_plan_from_labels(lab, hm, frames[:2])
    t0 = time.perf_counter()
    pipeline.crop_plan_from_labels(lab, hm, frames[:n_host])
    t_host = (time.perf_counter() - t0) / n_host
    print(f"frames {n_frames} candidates {len(c['frame_idx'])} crops {plan['crop_params'].shape[0]}")
    print(f"device plan (pack + H2D + kernel + compaction): {t_dev*1e3:.2f} ms  = {t_dev/n_frames*1e6:.2f} us/frame")
    print(f"ut_gen_crop_cameras call incl. output alloc: {t_k:.3f} ms")
    print(f"host per-frame path: {t_host*1e3:.3f} ms/frame  -> x{t_host/(t_dev/n_frames):.0f}")


if __name__ == "__main__":
    main()
