#!/usr/bin/env python3
"""Micro-benchmark of ut_backbone alone (for rocprofv3 --pmc / kernel-trace runs).
    python tools/bench_backbone.py [n_crops] [iters] [chunk]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from absolutetrack_amd import _native, arch, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = _native.HipEngine(synth.synthetic_state_dict(0), "cuda:0")
if len(sys.argv) > 3:
    eng.set_backbone_chunk(int(sys.argv[3]))
x = torch.rand(n, 96, 96, device="cuda:0")
out = torch.empty(n, 72, 6, 6, device="cuda:0")
eng.backbone(x, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    eng.backbone(x, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"backbone {n} crops: {dt*1e3:.3f} ms  {n/dt:.0f} crops/s  {n*arch.FLOPS_PER_CROP_BACKBONE/dt/1e12:.1f} TFLOP/s")
