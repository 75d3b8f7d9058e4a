"""Time the torch_data crop path (row f2) on the GPU against the oracle's numpy restatement of the reference.
usage: python tools/bench_torch_data.py [n_sequences]   (each sequence: 4 frames x 2 views of 480x636 u8)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from absolutetrack_amd import _native  # noqa: E402
from oracle import ref_torch_data as rt, scenarios  # noqa: E402  (CPU comparison leg only)


def main():
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
