"""A/B timing of the whole backbone (split-fp16 arithmetic, 4096 crops) under ut_set_resident_weights kinds, interleaved in one process.
    python tools/diag/resident_ab.py 1 6 [0 ...]        AB_ROUNDS=<n> (default 8), AB_CROPS=<n> (default 4096)"""
import os
import random
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from absolutetrack_amd import _native, synth  # noqa: E402

kinds = [int(a) for a in sys.argv[1:]] or [1, 6]
n = int(os.environ.get("AB_CROPS", "4096"))
eng = _native.HipEngine(synth.synthetic_state_dict(0), "cuda:0")
eng.set_conv_arithmetic("split_f16")
crops = torch.from_numpy(synth.synthetic_crops(64, seed=1)).to("cuda:0").repeat((n + 63) // 64, 1, 1)[:n].contiguous()
times = {k: [] for k in kinds}
random.seed(0)
for rnd in range(int(os.environ.get("AB_ROUNDS", "8"))):
    for k in random.sample(kinds, len(kinds)):
        eng.set_resident_weights(k)
        for _ in range(2):
            eng.backbone(crops)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng.backbone(crops)
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 3)
for k in kinds:
    t = times[k]
    print(f"ut_set_resident_weights({k}): backbone of {n} crops  median {statistics.median(t):7.3f} ms   min {min(t):7.3f} ms")
eng.set_resident_weights(1)
eng.poll_status()
eng.close()
