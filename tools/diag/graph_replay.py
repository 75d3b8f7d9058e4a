#!/usr/bin/env python3
"""Diagnostic: capture ONE batched step (ut_warp_backbone + ut_fuse_temporal_regress + ut_fk, deferred index checks) into a
hipGraph through torch.cuda.CUDAGraph, replay it twice and compare with the eager step.  Run ONCE under `timeout`.
    python tools/diag/graph_replay.py [frames] [conv]"""
import faulthandler
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from absolutetrack_amd import _native, pipeline, synth  # noqa: E402

faulthandler.dump_traceback_later(60, exit=True)
f = int(sys.argv[1]) if len(sys.argv) > 1 else 64
conv = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda", 0)
lab = pipeline.load_labels()
hm = pipeline.hand_model_from_labels(lab)
eng = _native.HipEngine(synth.synthetic_state_dict(0), dev)
eng.set_conv_arithmetic(conv)
g = torch.Generator(device=dev)
g.manual_seed(3)
src = torch.randint(0, 256, (f * 4, 480, 636), dtype=torch.uint8, device=dev, generator=g)
plan = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(lab, hm, range(f), dev).items()}
batch = pipeline.make_batch(plan, src, dev)
hot = pipeline.HotPath(eng, hm)
src_b = torch.randint(0, 256, (f * 4, 480, 636), dtype=torch.uint8, device=dev, generator=g)
src_a = src.clone()
want_a = hot.step(batch).clone()        # eager (also sizes every workspace: nothing allocates during capture)
batch.src.copy_(src_b)
want_b = hot.step(batch).clone()
hot.check()
assert not torch.equal(want_a, want_b)
torch.cuda.synchronize()
print("eager ok", flush=True)
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        rec = hot.step(batch)
print("captured", flush=True)
for i, (inp, want) in enumerate(((src_a, want_a), (src_b, want_b), (src_a, want_a), (src_b, want_b))):
    batch.src.copy_(inp)                # the graph reads the batch tensors in place
    rec.zero_()
    graph.replay()
    torch.cuda.synchronize()
    print(f"replay {i}: equal to the eager step on the same input = {bool(torch.equal(rec, want))}", flush=True)
hot.check()
print("done", flush=True)
