#!/usr/bin/env python3
"""Diagnostic (not part of the product): the fused layer1 BasicBlock kernel (conv_block32.hip) on synthetic data - error against
a float64 block on the first images, and interleaved timing of text-patched variants (timing-only ablations give wrong numbers).
    BLOCK_VARIANTS=a,b python tools/diag/block_ab.py [n_img]"""
import ctypes
import os
import statistics
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "absolutetrack_amd", "csrc")
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hw = 48
VARIANTS = {
    "nowait": [('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    if (tid == 0) slot_write(cur, grid + ticket);', '    if (tid == 0) slot_write(cur, grid + ticket);')],
    "nodma": [("        if (has_next && tap < B_MAXP) issue_piece(tap, n_row0, n_y0, n_x0, cur ^ 1);", "        if (has_next && tap < B_MAXP && p.n_img < 0) issue_piece(tap, n_row0, n_y0, n_x0, cur ^ 1);")],
    "nostore": [("        __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, (unsigned)(m * C + 8 * g4 + 4 * fh) * 4u, 0, 0);",
                 "        __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, p.n_img < 0 ? 0u : B_OOB, 0, 0);")],
    "noconvert": [("      for (int row = tid - 64 * U2; row < BP_PIX; row += 64 * (B_WAVES - U2)) convert_row(nreg, row);",
                   "      for (int row = tid - 64 * U2; row < BP_PIX && p.n_img < 0; row += 64 * (B_WAVES - U2)) convert_row(nreg, row);")],
}
variants = [v for v in os.environ.get("BLOCK_VARIANTS", "").split(",") if v]
ALT_SRC = os.environ.get("BLOCK_ALT_SRC")
W4_SRC = os.path.join(ROOT, "tools", "diag", "conv_block32w_experiment.hip")      # the 4-wave experiment, timed as "w4"


STAMPS = os.environ.get("BLOCK_STAMPS") == "1"      # w4 built with -DW4_STAMPS: phase stamps of every workgroup's 4th tile


def build(name, patches, alt=None):
    src = alt or os.path.join(CSRC, "conv_block32.hip")
    if patches:
        text = open(src).read()
        for old, new in patches:
            assert old in text, old
            text = text.replace(old, new)
        src = f"/tmp/conv_block32_{name}.hip"
        open(src, "w").write(text)
    so = f"/tmp/libblockab_{name}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-DBLOCK_W4", *(["-DW4_STAMPS"] if STAMPS else []), "-o", so, src,
                           W4_SRC, os.path.join(CSRC, "conv_split.hip"), os.path.join(CSRC, "conv_c64k.hip"),
                           os.path.join(ROOT, "tools", "diag", "block_entry.hip"), "-I", CSRC])
    return ctypes.CDLL(so)


libs = {"product": build("product", [])}
libs["w4"] = libs["product"]          # the same library: the 4-wave kernel of tools/diag/conv_block32w_experiment.hip
for v in variants:
    patches = []
    for part in v.split("+"):
        patches += VARIANTS[part]
    libs[v] = build(v.replace("+", "_"), patches)
if ALT_SRC:
    libs["alt"] = build("alt", [], alt=ALT_SRC)
dev = "cuda:0"
torch.manual_seed(0)
x = torch.relu(torch.randn(n_img, hw, hw, 32, device=dev) * 0.7 + 0.2)
ws = [torch.randn(32, 32, 3, 3) * (2.0 / (9 * 32)) ** 0.5 for _ in range(2)]
bs = [torch.randn(32) * 0.1 for _ in range(2)]


def packed(w):      # [128][288], k = tap * 32 + c
    wp = torch.zeros(128, 288)
    wp[:32] = w.permute(0, 2, 3, 1).reshape(32, 288)
    return wp


state = {}
for name, lib in libs.items():
    planes = []
    for i in range(2):
        sp = np.zeros(2 * 128 * 288, np.uint16)
        assert lib.block_pack(packed(ws[i]).numpy().ctypes.data_as(ctypes.c_void_p), i, sp.ctypes.data_as(ctypes.c_void_p)) == 0
        planes.append(torch.from_numpy(sp.view(np.int16)).to(dev))
    state[name] = planes
bias = [torch.cat([b, torch.zeros(96)]).to(dev) for b in bs]
in_max = x.max().reshape(1).view(torch.int32).clone()
out = torch.empty_like(x)


def run(name):
    lib, pl = libs[name], state[name]
    lib.block_run.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rc = lib.block_run(x.data_ptr(), out.data_ptr(), pl[0].data_ptr(), pl[1].data_ptr(), bias[0].data_ptr(), bias[1].data_ptr(),
                       float(bs[0].abs().max()) * 1.0001, in_max.data_ptr(), n_img, hw, int(name == "w4"))
    assert rc == 0, rc


nref = min(n_img, 4)
xr = x[:nref].permute(0, 3, 1, 2).double().cpu()
mid = torch.relu(torch.nn.functional.conv2d(xr, ws[0].double(), bs[0].double(), 1, 1))
ref = torch.relu(torch.nn.functional.conv2d(mid, ws[1].double(), bs[1].double(), 1, 1) + xr).permute(0, 2, 3, 1)
prod = None
for nm in ("product", "w4"):
    out.fill_(float("nan"))
    run(nm)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all(), nm
    print(f"{nm}: max |out - f64 block| over {nref} images = {float((out[:nref].double().cpu() - ref).abs().max()):.3e}  (|out| max {float(out.abs().max()):.2f})")
    if prod is None:
        prod = out.clone()
    else:
        print(f"{nm} vs product: max abs diff over all images {float((out - prod).abs().max()):.3e}")
if STAMPS:
    run("w4")
    buf = np.zeros(256 * 8, np.uint64)
    assert libs["w4"].block_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(256, 8)[:, :6].astype(np.int64)
    d = np.diff(st, axis=1)
    names = ["residual reads + convert (S1, S2)", "conv1 MFMA loop", "epilogue 1 + S3 + I writes + S4", "conv2 MFMA loop", "epilogue 2 (stores)"]
    print("w4 phase stamps, 4th tile of every workgroup (cycles: median / p10 / p90):")
    for i, nm in enumerate(names):
        print(f"   {nm:40s} {int(np.median(d[:, i])):7d} {int(np.percentile(d[:, i], 10)):7d} {int(np.percentile(d[:, i], 90)):7d}")
    print(f"   {'tile total':40s} {int(np.median(st[:, 5] - st[:, 0])):7d}")
    sys.exit(0)
flops = 2.0 * n_img * hw * hw * 32 * 288 * 2
times = {k: [] for k in libs}
import random
random.seed(1)
for rnd in range(int(os.environ.get("AB_ROUNDS", "10"))):
    for name in random.sample(list(libs), len(libs)):      # a new order every round, a lead-in per case: position effects average out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            run(name)
        e0.record()
        for _ in range(4):
            run(name)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 4)
for name in libs:
    t = times[name]
    med, mn = statistics.median(t), min(t)
    print(f"{name:22s} median {med*1e3:8.1f} us ({flops/med/1e9:6.1f} TF-equivalent)   min {mn*1e3:8.1f} us ({flops/mn/1e9:6.1f})")
