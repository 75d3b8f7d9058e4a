"""How much accuracy would a split-bf16 MFMA convolution cost?  (CPU study, TEST INFRASTRUCTURE ONLY.)

The product's convolutions run exact fp32 MFMA (v_mfma_f32_32x32x2_f32, 256 flop/cycle/CU); the bf16 MFMA of gfx950
issues 16x the flops per cycle.  An fp32 value splits into bf16 pieces  x = x0 + x1 (+ x2), each product of two pieces
is exact in fp32, and the MFMA accumulates in fp32 - so a "bf16xN" convolution is N bf16 MFMA products per fp32 one:
    1 product   x0*w0                              (plain bf16 inputs)
    3 products  x0*w0 + x0*w1 + x1*w0              (~16 mantissa bits kept)
    6 products  + x1*w1 + x0*w2 + x2*w0            (~24 bits: fp32-equivalent up to accumulation order)
This script runs the oracle's network with every convolution of cin >= 32 replaced by each variant (values rounded to
bf16 and multiplied in fp32: bit-for-bit what the MFMA would accumulate, up to summation order) and reports the error
against the unmodified fp32 oracle at the path's outputs, next to north_star's tolerance (1e-4 rad).

    python -m oracle.studies.split_precision [--frames 4]
"""
import argparse
import json

import numpy as np
import torch
import torch.nn.functional as F

from absolutetrack_amd import pipeline, synth
from oracle import checks, ref_model


def _split(x: torch.Tensor, n: int, dtype=torch.bfloat16):
    parts, r = [], x
    for _ in range(n):
        p = r.to(dtype).float()
        parts.append(p)
        r = r - p
    return parts


class _SplitF:
    """Stands in for torch.nn.functional inside oracle.ref_model: conv2d goes through the split, the rest passes through."""

    def __init__(self, products: int):
        self.products = products

    def __getattr__(self, name):
        return getattr(F, name)

    def conv2d(self, x, w, bias=None, stride=1, padding=0, *a, **k):
        if self.products == 0 or x.shape[1] < 32:
            return F.conv2d(x, w, bias, stride, padding, *a, **k)
        if self.products in (-3, -4):
            # fp16 pieces (11 significand bits each, two per operand): x0 w0 + x0 w1 + x1 w0 (+ x1 w1); the weights are
            # pre-scaled by a power of two so that their second piece stays a normal fp16 number, the activations are not
            sc = 2.0 ** (14 - int(torch.ceil(torch.log2(w.abs().max()))))
            xs, ws = _split(x, 2, torch.float16), _split(w * sc, 2, torch.float16)
            assert all(torch.isfinite(t).all() for t in xs + ws)
            pairs = [(1, 1)] * (self.products == -4) + [(0, 1), (1, 0), (0, 0)]
            y = None
            for i, j in pairs:
                t = F.conv2d(xs[i], ws[j], None, stride, padding, *a, **k)
                y = t if y is None else y + t
            y = y * (1.0 / sc)
            return y if bias is None else y + bias.view(1, -1, 1, 1)
        n = {1: 1, 3: 2, 6: 3}[self.products]
        xs, ws = _split(x, n), _split(w, n)
        pairs = {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]}[self.products]
        y = None
        for i, j in reversed(pairs):                       # small terms first
            t = F.conv2d(xs[i], ws[j], None, stride, padding, *a, **k)
            y = t if y is None else y + t
        return y if bias is None else y + bias.view(1, -1, 1, 1)


def run(n_frames: int, known: bool):
    lab = pipeline.load_labels()
    hm_np = {k[3:]: v for k, v in lab.items() if k.startswith("hm.")}
    sd = synth.synthetic_state_dict(0)
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (n_frames, 4, 480, 636), dtype=np.uint8)
    out = {}
    base = None
    for products in (0, 1, 3, 6, -3, -4):
        ref_model.F = _SplitF(products)
        try:
            o = checks.oracle_frames(sd, lab, hm_np, range(n_frames), frames, known=known)
        finally:
            ref_model.F = F
        if base is None:
            base = o
            continue
        out[f"bf16x{products}" if products > 0 else f"fp16x{-products}"] = {
            "max_abs_joint_angle_rad": float(np.abs(o["joint_angles"] - base["joint_angles"]).max()),
            "max_abs_wrist_xf": float(np.abs(o["wrist_xfs"] - base["wrist_xfs"]).max()),
            "max_keypoint_mm": float(np.linalg.norm(o["keypoints_mm"] - base["keypoints_mm"], axis=-1).max()),
        }
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    a = ap.parse_args()
    torch.set_num_threads(8)
    res = {"known": run(a.frames, True), "unknown": run(a.frames, False), "frames": a.frames,
           "tolerance_rad": 1e-4, "weights": "synthetic seed 0"}
    print(json.dumps(res, indent=1))
