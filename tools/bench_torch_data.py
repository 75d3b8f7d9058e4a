"""Time the torch_data crop path (row f2) on the GPU against the oracle's numpy restatement of the reference.
usage: python tools/bench_torch_data.py [n_sequences]   (each sequence: 4 frames x 2 views of 480x636 u8)
       python tools/bench_torch_data.py --eval-batch [batch_size] [--conv fp32|split_f16]
           the model side of run_inference_torch_data.py:88-135: a collated [bs, 4, 2, 96, 96] batch through _eval_batch (temporal
           memory engaged from the second step), timed in the exact-fp32 and in the chosen arithmetic, with their largest difference"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from absolutetrack_amd import _native  # noqa: E402
from oracle import ref_torch_data as rt, scenarios  # noqa: E402  (CPU comparison leg only)


def eval_batch_main(argv):
    from absolutetrack_amd import bundles, pipeline, synth, torch_data as td
    from absolutetrack_amd.model import UmeTrackModel
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_gpu_torch_data import _raw_sample
    conv = "split_f16"
    if "--conv" in argv:
        i = argv.index("--conv")
        conv = argv[i + 1]
        del argv[i: i + 2]
    bs = int(argv[0]) if argv else 1024
    dev = torch.device("cuda", 0)
    lab = pipeline.load_labels()
    pairs = [td.prepare_inputs_targets(_raw_sample(h, lab), (96, 96)) for h in (0, 1)]
    rep = lambda t: t.repeat(bs // 2, *([1] * (t.ndim - 1))) if isinstance(t, torch.Tensor) else t
    model_input = bundles.map_fields(rep, bundles.collate([p[0] for p in pairs]), only_type=torch.Tensor)
    model_target = bundles.map_fields(rep, bundles.collate([p[1] for p in pairs]), only_type=torch.Tensor)
    # make the replicas differ: a per-sequence brightness change of the crops
    gain = 0.6 + 0.4 * torch.rand(model_input.left_images.shape[0], 1, 1, 1, 1, generator=torch.Generator().manual_seed(3))
    model_input.left_images = torch.floor(model_input.left_images * gain * 255.0) / 255.0
    model = UmeTrackModel(synth.synthetic_state_dict(0))
    model.eval()
    model.to(dev)
    seq = model_input.left_images.shape[1]
    res = {}
    for mode in ("fp32", conv):
        model.engine.set_conv_arithmetic(mode)
        model.reset_temporal_memory()
        _gt, kp = td.eval_batch_keypoints(model, model_input, model_target, "multiv", True, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.engine.reset_memory()
            td._eval_batch(model, model_input, model_target, "multiv", True, dev)
        torch.cuda.synchronize()
        res[mode] = (kp, (time.perf_counter() - t0) / (3 * seq))
        model.engine.poll_status()
    print(f"_eval_batch: {bs} sequences x {seq} steps x 2 views (run_inference_torch_data.py:88-135; host tensors in, per-step read-back as in the reference)")
    for mode, (_kp, dt) in res.items():
        print(f"  {mode:10s} {dt * 1e3:8.2f} ms per time step = {bs / dt:9.0f} hand-frames/s")
    if conv != "fp32":
        d = float((res[conv][0] - res["fp32"][0]).abs().max()) * 1000.0
        print(f"  max |{conv} - fp32| over all {bs * seq} keypoint sets: {d:.3e} mm (tolerance 1e-3 mm)")


def main():
    if "--eval-batch" in sys.argv:
        argv = [a for a in sys.argv[1:] if a != "--eval-batch"]
        return eval_batch_main(argv)
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    c = scenarios.torch_data_case(0, h=480, w=636)
    f, v = c["images"].shape[:2]
    dev = "cuda:0"
    rep = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).repeat(n_seq, *([1] * (a.ndim - 1)))
    img, ext, intr, pts = rep(c["images"]), rep(c["extrinsics"]), rep(c["intrinsics"]), rep(c["crop_points"])
    hand = torch.zeros(img.shape[0], dtype=torch.int64, device=dev)
    n = img.shape[0] * v
    out = torch.empty(n, 96, 96, device=dev)

    def run():
        m = _native.gen_crop_matrices(ext, intr, pts, hand)
        _native.resample_homography(img.reshape(-1, 480, 636), m["resample_xf"].reshape(-1, 4, 4), (96, 96), out=out)
        return m
    run()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(10):
        m = _native.gen_crop_matrices(ext, intr, pts, hand)
    e[1].record()
    for _ in range(10):
        _native.resample_homography(img.reshape(-1, 480, 636), m["resample_xf"].reshape(-1, 4, 4), (96, 96), out=out)
    e[2].record()
    torch.cuda.synchronize()
    t_m, t_r = e[0].elapsed_time(e[1]) / 10, e[1].elapsed_time(e[2]) / 10
    t0 = time.perf_counter()
    rt.perspective_crop_images(c["images"], c["extrinsics"], c["intrinsics"], c["crop_points"], 0, (96, 96))
    t_cpu = (time.perf_counter() - t0) / (f * v)
    print(f"{n} crops from {img.shape[0]} frames x {v} views (480x636 u8)")
    print(f"ut_gen_crop_matrices: {t_m:.3f} ms   ut_resample_homography: {t_r:.3f} ms "
          f"({n * 96 * 96 * 4 / t_r / 1e6:.1f} GB/s of crop output)")
    print(f"GPU {1e3 * (t_m + t_r) / n:.3f} us/crop   oracle numpy {t_cpu * 1e6:.0f} us/crop   x{t_cpu * 1e3 / ((t_m + t_r) / n):.0f}")


if __name__ == "__main__":
    main()
