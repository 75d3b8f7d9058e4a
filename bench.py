#!/usr/bin/env python3
"""Headline benchmark: hand-frames/s of the UmeTrack per-frame hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One "step" = one pass of the whole hot path (fisheye->pinhole resample of every crop, backbone,
fusion/temporal/regressor head, decode, forward kinematics) over one shard of synthetic frames:
4 fisheye cameras x 2 hands per frame, 2 selected views per hand (BASELINE.json configs[2] shapes;
cameras / poses / hand model from sample_data/recording_00.json, frame i mod 369; u8 noise images;
seeded synthetic weights).  Inputs are resident in HBM before the timed region.  Frames shard
contiguously across ranks (weak scaling: --frames-per-gpu each) and the packed per-hand records are
all-gathered over RCCL at the end of every step.

Extra JSON objects: `roofline` (the implicit-GEMM convolution kernel, hipEvent-timed per launch in a
separate profiling pass on the launch stream) and `cpu_baseline` (the CPU oracle's restatement of the
same path on a bounded sample, rank 0, N=1 only).  The same checker leg also drives all 369 label frames of
recording_00 as one sequence (temporal memory engaged) through the drop-in HandTracker and the oracle and
reports `mpjpe_delta_mm` (BASELINE.json: within 0.05 mm); --parity-frames 0 skips it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense fp16/bf16 MFMA; the split kernels spend 3 fp16 MFMA flops per flop


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-gpu", type=int, default=1024)
    ap.add_argument("--mode", choices=["known", "unknown"], default="known")
    ap.add_argument("--chunk", type=int, default=0, help="crops per backbone pass (0 = library default)")
    ap.add_argument("--cpu-frames", type=int, default=768, help="label frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--parity-frames", type=int, default=369,
                    help="label frames of recording_00 run as a sequence against the oracle for mpjpe_delta_mm (0 = skip)")
    ap.add_argument("--lanes", type=int, default=1, choices=[1, 2],
                    help="backbone lanes: 2 = two half-batches on two internal streams (ut_set_backbone_lanes; +0.3 %%, "
                         "but concurrent launches make per-kernel durations in a rocprof trace overlap, so the default "
                         "keeps one lane and the trace comparable with the roofline leg)")
    ap.add_argument("--conv", choices=["fp32", "split_f16"], default="split_f16",
                    help="arithmetic of the batched backbone convolutions (ut_set_conv_arithmetic): exact fp32 matrix "
                         "instructions, or two-piece fp16 splits of both operands on the fp16 matrix cores (fp32-level error)")
    ap.add_argument("--no-fp32-mode", action="store_true",
                    help="with --conv split_f16: skip the exact-fp32 timing and the record comparison against it (profiling runs)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for N>1 (nccl = RCCL over xGMI; gloo only to rehearse the multi-process "
                         "control flow on a box with fewer GPUs than ranks: ranks then share devices)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step's launches from one captured hipGraph")
    ap.add_argument("--cropgen-in-step", action="store_true",
                    help="also regenerate the crop cameras from the label poses inside every step (SURVEY 8 f1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from absolutetrack_amd import _native, arch, pipeline, synth

    known = args.mode == "known"
    sd = synth.synthetic_state_dict(0)
    lab = pipeline.load_labels()
    hm = pipeline.hand_model_from_labels(lab)
    eng = _native.HipEngine(sd, device)
    if args.chunk:
        eng.set_backbone_chunk(args.chunk)
    eng.set_backbone_lanes(args.lanes)
    eng.set_conv_arithmetic(args.conv)

    f_local = args.frames_per_gpu
    lo, hi = pipeline.shard_frames(f_local * world, rank, world)
    plan = {k: v.cpu().numpy() for k, v in pipeline.crop_plan_on_device(lab, hm, range(lo, hi), device).items()}
    planner = pipeline.DeviceCropPlanner(lab, hm, range(lo, hi), device) if args.cropgen_in_step else None
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    src = torch.randint(0, 256, (f_local * 4, 480, 636), dtype=torch.uint8, device=device, generator=gen)
    batch = pipeline.make_batch(plan, src, device)
    hot = pipeline.HotPath(eng, hm, known_skeleton=known)
    s_local, n_local = batch.n_samples, batch.n_crops

    def barrier():
        if world > 1:
            dist.barrier()

    # every rank holds an equal frame block here; check it once so that the per-step gather can skip the count exchange
    # (whose read-back would stop the host from running ahead of the GPU)
    equal = True
    if world > 1:
        c = torch.tensor([s_local, -s_local], dtype=torch.int64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(c, op=dist.ReduceOp.MAX)
        equal = int(c[0]) == -int(c[1])

    # --graph: the launches of one step (resample + backbone, head, FK: ~60 kernels) captured once into a hipGraph and replayed per
    # step - same kernels, same order, same stream; the gather stays eager
    graph_state = {}

    def one_step():
        if planner is not None:
            planner.refresh(batch)
        if args.graph and planner is None:
            if "g" not in graph_state:
                hot.step(batch)                       # (workspace and modes settled by an eager step first)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream(device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side):
                        graph_state["rec"] = hot.step(batch)
                torch.cuda.current_stream().wait_stream(side)
                graph_state["g"] = g
            graph_state["g"].replay()
            rec = graph_state["rec"]
        else:
            rec = hot.step(batch)
        return pipeline.gather_records(rec, world, equal_counts=equal)

    def timed_run():
        out = None
        for _ in range(args.warmup):
            out = one_step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = one_step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    dt, out = timed_run()
    hot.check()            # deferred index checks of every step above
    assert out.shape == (s_local * world, pipeline.RECORD)
    finite = bool(torch.isfinite(out).all().item())
    if planner is not None and not bool(planner.ok.item()):
        raise SystemExit("crop-camera generation inside the step produced a candidate without two views")

    total_hf = s_local * world * args.steps
    value = total_hf / dt
    flops_hf = arch.FLOPS_PER_HANDFRAME_KNOWN if known else arch.FLOPS_PER_HANDFRAME_UNKNOWN

    # split-fp16 mode: the same K steps timed with the exact-fp32 convolutions (every rank, same barriers), and the records
    # of the timed workload against that mode's on the same batch
    split_check, fp32_mode = None, None
    if args.conv != "fp32" and not args.no_fp32_mode:
        rec_split = hot.step(batch).clone()
        eng.set_conv_arithmetic("fp32")
        dt32, _ = timed_run()
        rec_fp32 = hot.step(batch).clone()
        eng.set_conv_arithmetic(args.conv)
        hot.check()
        fp32_mode = {"value": round(total_hf / dt32, 1), "unit": "hand-frames/s", "ms_per_step": round(dt32 / args.steps * 1e3, 3),
                     "what": "the same steps with ut_set_conv_arithmetic(UT_CONV_FP32): every convolution on v_mfma_f32_32x32x2_f32"}
        split_check = {"against": "the same step with UT_CONV_FP32 (itself pinned to the oracle by tests/ and parity_recording_00)",
                       "hand_frames": int(rec_split.shape[0]),
                       "max_joint_angle_diff_rad": float((rec_split[:, :22] - rec_fp32[:, :22]).abs().max()),
                       "max_keypoint_diff_mm": float((rec_split[:, 60:] - rec_fp32[:, 60:]).reshape(rec_split.shape[0], -1, 3)
                                                     .norm(dim=-1).max()),
                       "tolerance": "BASELINE.json north_star: 1e-4 rad / 1e-3 mm"}

    roofline = None
    # (split arithmetic requested; with the roofline pass: requested AND its launches were seen)
    split_kind = args.conv != "fp32"
    if not args.no_roofline:
        # separate profiling pass: hipEvents around every conv_igemm launch on the launch stream
        eng.profile_begin()
        n_prof = 2
        for _ in range(n_prof):
            hot.step(batch)
        kinds = eng.profile_end_by_kind()
        split_kind = args.conv != "fp32" and kinds[1][1] > 0
        ms, launches, flops = kinds[1] if split_kind else kinds[0]
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM bytes per launch cannot be measured by this process: they come from separate rocprofv3 --pmc passes of
        # this same command (FETCH_SIZE and WRITE_SIZE cannot share a pass), summarised by tools/pmc_traffic.py into
        # profiles/ and quoted here with their source; null when no summary matches this workload
        traffic, traffic_source, traffic_step = None, None, None
        prof_dir = os.path.join(ROOT, "profiles")
        cands = sorted(f for f in (os.listdir(prof_dir) if os.path.isdir(prof_dir) else []) if f.endswith("_conv_traffic.json"))
        cands = [f for f in cands
                 if json.load(open(os.path.join(prof_dir, f))).get("conv_arithmetic", "fp32") == ("split_f16" if split_kind else "fp32")]
        if cands and f_local == 1024 and known:
            tj = json.load(open(os.path.join(prof_dir, cands[-1])))
            traffic = tj.get("traffic_bytes_per_launch")
            traffic_step = {"all_kernels_bytes_per_step": tj.get("step_traffic_bytes_all_kernels"),
                            "ratio_to_algorithmic": tj.get("ratio_to_algorithmic"),
                            "by_kernel_bytes_per_step": dict(list((tj.get("step_traffic_bytes_by_kernel") or {}).items())[:8])}
            traffic_source = (f"profiles/{cands[-1]}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `{tj.get('label', '')}` "
                              "(an earlier run of this command, not this process); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024")
        peak = PEAK_F16_MFMA_TFLOPS / 3.0 if split_kind else PEAK_FP32_MFMA_TFLOPS
        roofline = {"bound": "mfma",
                    "kernel": ("conv_w4_kernel (stride-1 3x3 of layer2 .. layer4 and of the pose regressor, and the stride-2 entries of layer3 / layer4 as phase planes: whole-map tiles) + conv_c32s2_kernel (layer2 entry, 3x3 / 2 + shortcut) + conv_block32_kernel (layer1, one launch per BasicBlock): "
                               "two-piece fp16 splits, 3 products per k on v_mfma_f32_32x32x16_f16" if split_kind else
                               "conv_igemm_kernel (all instantiations) + conv3x3_c32_patch_kernel (layer1)"),
                    "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "peak_source": ("dense fp16 MFMA peak 2500 TFLOP/s / 3 products per algorithmic flop" if split_kind else
                                    "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_source": traffic_source, "traffic_whole_step": traffic_step,
                    "launches_per_step": launches // n_prof, "avg_launch_ms": round(ms / max(launches, 1), 5),
                    "flops_per_launch_avg": flops / max(launches, 1),
                    "whole_step_tflops": round(value * flops_hf / 1e12, 3)}
        if split_kind and kinds[0][0] > 0:
            ms0, l0, f0 = kinds[0]
            roofline["fp32_kernels_in_this_mode"] = {
                "what": "1x1 shortcut convolutions, projection and head: still conv_igemm_kernel on the fp32 matrix instructions",
                "launches_per_step": l0 // n_prof, "ms_per_step": round(ms0 / n_prof, 4),
                "achieved": round(f0 / (ms0 * 1e-3) / 1e12, 3), "peak": PEAK_FP32_MFMA_TFLOPS}

    # the same profiling pass with the exact-fp32 convolutions, for the fp32-MFMA roofline beside the split one
    roofline_fp32 = None
    if roofline is not None and args.conv != "fp32" and not args.no_fp32_mode:
        eng.set_conv_arithmetic("fp32")
        eng.profile_begin()
        for _ in range(2):
            hot.step(batch)
        ms32, l32, f32_ = eng.profile_end_by_kind()[0]
        eng.set_conv_arithmetic(args.conv)
        hot.check()
        a32 = f32_ / (ms32 * 1e-3) / 1e12 if ms32 > 0 else 0.0
        roofline_fp32 = {"bound": "mfma", "kernel": "conv_igemm_kernel (all instantiations) + conv3x3_c32_patch_kernel (layer1)",
                         "achieved": round(a32, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(a32 / PEAK_FP32_MFMA_TFLOPS, 4), "launches_per_step": l32 // 2,
                         "avg_launch_ms": round(ms32 / max(l32, 1), 5)}

    cpu, batched = None, None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle import checks
        # the GPU box gives one GPU a 16-CPU share although os.cpu_count() reports the whole host
        threads = min(os.cpu_count() or 1, 16)
        r = checks.time_oracle(sd, args.cpu_frames, threads)
        # the batched path, raw images -> keypoints, against the oracle for the frames it has just computed, in both
        # arithmetics: with identical crop cameras, with each side's own cameras (cv2 remap), and in float-remap mode
        batched = checks.batched_parity(sd, r, str(device)) if known else None
        cpu = {"value": round(r["hand_frames"] / r["seconds"], 2), "unit": "hand-frames/s", "cores": threads,
               "kind": "port",
               "sample": f"{args.cpu_frames} label frames ({r['hand_frames']} hand-frames) of the same workload, "
                         f"oracle resample+network+FK, {r['seconds']:.1f} s"}

    parity = None
    if rank == 0 and world == 1 and args.parity_frames > 0:
        from oracle import checks      # checker leg, like cpu_baseline above: never inside the timed region
        eng.close()
        r = checks.run_recording00(sd, str(device), known=known, n_frames=args.parity_frames)
        parity = {k: r[k] for k in ("mode", "frames", "hand_frames", "mpjpe_build_mm", "mpjpe_oracle_mm", "mpjpe_delta_mm",
                                    "max_joint_angle_err_rad", "max_keypoint_err_mm", "oracle_cpu_seconds") if k in r}
        if not known:
            parity.update({k: r[k] for k in ("scale_mean_build", "scale_mean_oracle", "scale_mean_abs_diff")})

    # what the process group saw: backend, world size and every rank's device (so that an N>1 record shows N ranks on N GPUs)
    me = {"rank": rank, "device": torch.cuda.get_device_name(device), "index": dev_index}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    if rank == 0:
        line = {
            "metric": "hand-frames/sec", "value": round(value, 1), "unit": "hand-frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.conv == "fp32" else "f32 (backbone products as two fp16 pieces per operand, fp32 accumulation)",
            "data": "synthetic",
            "config": {"workload": "4 fisheye cameras x 2 hands per frame, 2 views per hand, 96x96 crops, "
                                   f"{'known' if known else 'unknown'}-skeleton path, full hot path "
                                   "(resample+backbone+head+FK) + all-gather of records"
                                   + (", crop cameras regenerated in the step" if planner is not None else ""),
                       "frames_per_gpu": f_local, "hand_frames_per_step": s_local * world,
                       "crops_per_step": n_local * world, "src_image": "480x636 u8 x 4 cameras",
                       "parallelism": f"frame-shard x{world}" + ("" if args.backend == "nccl" else " (gloo rehearsal)"),
                       "world": world, "backend": (dist.get_backend() if world > 1 else "none (single process)"),
                       "collective": "one all_gather_into_tensor of the [S_local,123] records per step" if world > 1 else None,
                       "ranks": ranks,
                       "outputs_finite": finite},
            "conv_arithmetic": args.conv,
            "split_scale": ("calibrated: fixed per-tensor powers of two from the built-in 64-crop calibration pass, 2^4 headroom, range-guarded"
                            if split_kind else None),
            "split_f16_check": split_check, "exact_fp32_mode": fp32_mode,
            "parity_batched_vs_oracle": batched,
            "roofline": roofline, "roofline_exact_fp32_mode": roofline_fp32, "cpu_baseline": cpu,
            "mpjpe_delta_mm": None if parity is None else parity["mpjpe_delta_mm"], "parity_recording_00": parity,
        }
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
