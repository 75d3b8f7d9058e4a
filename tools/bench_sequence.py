"""Sequence-mode run of the batched hot path (SURVEY.md section 8 d: "one sequence-mode run, seq len 8, use_memory=True
after step 0"): S hand-sequences advance together, every step warps the slot's temporal memory by
cur_ext * prev_ext^-1 and feeds it back (lib/models/temporal.py:51-139); checks that the state is engaged and
times the steps, in the exact-fp32 arithmetic and (--conv split_f16, the default) in the split-fp16 arithmetic of the backbone, with
the largest difference between the two over all steps (the recurrence feeds each step's features into the next).
     python tools/bench_sequence.py [frames_per_step] [seq_len] [--conv fp32|split_f16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from absolutetrack_amd import _native, pipeline, synth  # noqa: E402


def main():
    argv = list(sys.argv[1:])
    conv = "split_f16"
    if "--conv" in argv:
        i = argv.index("--conv")
        conv = argv[i + 1]
        del argv[i: i + 2]
    f = int(argv[0]) if len(argv) > 0 else 1024
    seq = int(argv[1]) if len(argv) > 1 else 8
    dev = torch.device("cuda", 0)
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    eng = _native.HipEngine(synth.synthetic_state_dict(0), dev)
    hot = pipeline.HotPath(eng, hm)
    # step t of sequence i shows label frame (i + t): neighbouring label frames = a moving hand
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    src = torch.randint(0, 256, (f * 4, 480, 636), dtype=torch.uint8, device=dev, generator=gen)
    batches = []
    for t in range(seq):
        plan = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(lab, hm, [i + t for i in range(f)], dev).items()}
        b = pipeline.make_batch(plan, src, dev)
        # this process owns the whole sequence block (pipeline.shard_sequences(f, 0, 1)): slots stay local, memory from step 1
        b.memory_idx, b.use_memory, b.n_slots = pipeline.sequence_step_descriptors(b.hand_idx, first_step=t == 0)
        batches.append(b)
    s = batches[0].n_samples
    assert all(b.n_samples == s for b in batches)

    def run():
        eng.reset_memory()
        return [hot.step(b).clone() for b in batches]
    out32 = run()                     # exact fp32 (the handle's default)
    torch.cuda.synchronize()
    eng.set_conv_arithmetic(conv)
    out = run()
    torch.cuda.synchronize()
    hot.check()
    d_ang = max(float((a[:, :22] - b[:, :22]).abs().max()) for a, b in zip(out, out32))
    d_kp = max(float((a[:, 60:] - b[:, 60:]).abs().max()) for a, b in zip(out, out32))
    # the memory matters: step 1 with memory differs from step 1 started cold
    eng.reset_memory()
    cold = batches[1]
    keep = cold.use_memory.clone()
    cold.use_memory = torch.zeros_like(keep)
    cold_out = hot.step(cold).clone()
    cold.use_memory = keep
    d = float((cold_out[:, :22] - out[1][:, :22]).abs().max())
    assert d > 1e-6, "temporal memory had no effect"
    mem, _ext = eng.get_memory()
    assert mem.shape[0] == s and float(mem.abs().max()) > 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * seq)
    print(f"{s} hand-sequences x {seq} steps, memory engaged from step 1 (max joint-angle change vs cold start {d:.3e} rad), backbone arithmetic {conv}")
    print(f"{dt * 1e3:.2f} ms per step = {s / dt:.0f} hand-frames/s")
    print(f"max over the {seq} steps of |{conv} - fp32|: joint angles {d_ang:.3e} rad, keypoints {d_kp:.3e} mm (tolerance 1e-4 rad / 1e-3 mm)")
    if conv != "fp32":
        eng.set_conv_arithmetic("fp32")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / (reps * seq)
        print(f"fp32 in the same process: {dt32 * 1e3:.2f} ms per step = {s / dt32:.0f} hand-frames/s")


if __name__ == "__main__":
    main()
