#!/usr/bin/env python3
"""Diagnostic (not part of the product): layer2's stride-2 entry (conv_c32s2.hip: 3x3 / 2 + 1x1 / 2 shortcut, 32 -> 64 channels, one
launch) on its own - both outputs against float64 convolutions, timing, and (S2_STAMPS=<wave 0..7>) phase stamps patched into a COPY
of the kernel (the product source carries no timing code).
    python tools/diag/s2_ab.py [n_img]          S2_ALT_SRCS=<path>,... other versions of conv_c32s2.hip timed beside the product"""
import ctypes
import os
import random
import statistics
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "absolutetrack_amd", "csrc")
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = W = 48

STAMP = ('if (k == 3) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(st_[I]) :: "memory"); '
         '__builtin_amdgcn_sched_barrier(0); }')


def stamp_patches(wave):
    st = lambda i: STAMP.replace("[I]", f"[{i}]")
    return [
        ("    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);\n", ""),
        ("  f32x16s acc, accd;\n  for (int k = 0; k < my_tiles; ++k) {\n", "  f32x16s acc, accd;\n  unsigned long long st_[8];\n  for (int k = 0; k < my_tiles; ++k) {\n    " + st(0) + "\n"),
        ("    // the next tile's patch: landed (every wave counted its pieces in), split by all;", "    " + st(1) + "\n    // the next tile's patch: landed (every wave counted its pieces in), split by all;"),
        ("      S_AWAIT(cnt_addr, 8 * (k + 1))\n", "      S_AWAIT(cnt_addr, 8 * (k + 1))\n      " + st(2) + "\n"),
        # inside the MFMA loop: around the wait for my own pieces (k-step 14); stamps 6 and 7
        ("      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");      /* the splitting waves start later: measured + 8 %) */                \\\n",
         "      " + st(6) + " \\\n      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); \\\n      " + st(7) + " \\\n"),
        ("    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    __builtin_amdgcn_s_barrier();\n    if (mfma_wave && wave < 4) S_EPILOGUE();\n",
         "    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    " + st(3) + "\n    __builtin_amdgcn_s_barrier();\n    " + st(4) + "\n    if (mfma_wave && wave < 4) S_EPILOGUE();\n    " + st(5) +
         "\n    if (k == 3 && tid == %d && p.status) {\n      unsigned long long* d = reinterpret_cast<unsigned long long*>(p.status) + blockIdx.x * 8;\n      for (int i = 0; i < 8; ++i) d[i] = st_[i];\n    }\n" % (64 * wave)),
    ]


def build(name, src=None, patches=()):
    src = src or os.path.join(CSRC, "conv_c32s2.hip")
    if patches:
        text = open(src).read()
        for old, new in patches:
            assert old in text, old
            text = text.replace(old, new, 1)
        src = f"/tmp/conv_c32s2_{name}.hip"
        open(src, "w").write(text)
    so = f"/tmp/libs2ab_{name}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-o", so, src,
                           os.path.join(CSRC, "conv_split.hip"), os.path.join(CSRC, "conv_c64k.hip"),
                           os.path.join(ROOT, "tools", "diag", "s2_entry.hip"), "-I", CSRC])
    return ctypes.CDLL(so)


stamp_wave = os.environ.get("S2_STAMPS")
lib = build("stamps", patches=stamp_patches(int(stamp_wave))) if stamp_wave else build("product")
alts = {os.path.splitext(os.path.basename(q))[0]: build("alt_" + os.path.splitext(os.path.basename(q))[0], src=q)
        for q in os.environ.get("S2_ALT_SRCS", "").split(",") if q}
dev = "cuda:0"
torch.manual_seed(0)
x = torch.rand(n_img, H, W, 32, device=dev) * 2 - 0.5
w1 = torch.randn(64, 32, 3, 3) * (2.0 / (9 * 64)) ** 0.5
wd = torch.randn(64, 32, 1, 1) * (2.0 / 64) ** 0.5
b1, bd = torch.randn(64) * 0.1, torch.randn(64) * 0.1
# packed k order (channel slice of 32, tap, channel in slice); cout padded to 128
wp1 = torch.zeros(128, 288)
wp1[:64] = w1.permute(0, 2, 3, 1).reshape(64, 288)
wpd = torch.zeros(128, 32)
wpd[:64] = wd.reshape(64, 32)
sp1, spd = np.zeros(2 * 128 * 288, np.uint16), np.zeros(2 * 128 * 32, np.uint16)
vp = ctypes.c_void_p
for l in [lib] + list(alts.values()):
    assert l.s2_pack(0, wp1.numpy().ctypes.data_as(vp), 128, 288, sp1.ctypes.data_as(vp)) == 0
    assert l.s2_pack(1, wpd.numpy().ctypes.data_as(vp), 128, 32, spd.ctypes.data_as(vp)) == 0
s1_d, sd_d = torch.from_numpy(sp1.view(np.int16)).to(dev), torch.from_numpy(spd.view(np.int16)).to(dev)
b1_d, bd_d = torch.cat([b1, torch.zeros(64)]).to(dev), torch.cat([bd, torch.zeros(64)]).to(dev)
in_max = x.abs().max().reshape(1).view(torch.int32).clone()
out_max = torch.zeros(1, dtype=torch.int32, device=dev)
o1 = torch.empty(n_img, 24, 24, 64, device=dev)
o2 = torch.empty_like(o1)


def run(l=lib):
    rc = l.s2_diag(vp(x.data_ptr()), vp(s1_d.data_ptr()), vp(sd_d.data_ptr()), vp(b1_d.data_ptr()), vp(bd_d.data_ptr()), vp(o1.data_ptr()),
                   vp(o2.data_ptr()), n_img, H, W, vp(in_max.data_ptr()), vp(out_max.data_ptr()), 1 if stamp_wave else 0)
    assert rc == 0, rc


o1.fill_(float("nan")); o2.fill_(float("nan"))
run()
torch.cuda.synchronize()
nref = min(n_img, 6)
xr = x[:nref].permute(0, 3, 1, 2).double().cpu()
r1 = torch.relu(torch.nn.functional.conv2d(xr, w1.double(), b1.double(), 2, 1)).permute(0, 2, 3, 1)
r2 = torch.nn.functional.conv2d(xr, wd.double(), bd.double(), 2, 0).permute(0, 2, 3, 1)
assert torch.isfinite(o1).all() and torch.isfinite(o2).all()
print(f"3x3 / 2  max |out - f64| over {nref} images = {float((o1[:nref].double().cpu() - r1).abs().max()):.3e}   (|out| max {float(o1.abs().max()):.2f}, max word "
      f"{float(out_max.view(torch.float32)):.4f})")
print(f"shortcut max |out - f64| over {nref} images = {float((o2[:nref].double().cpu() - r2).abs().max()):.3e}   (|out| max {float(o2.abs().max()):.2f})")
if stamp_wave:
    buf = np.zeros(256 * 8, np.uint64)
    assert lib.s2_stamps(buf.ctypes.data_as(vp)) == 0
    full = buf.reshape(256, 8).astype(np.int64)
    print(f"   wave {stamp_wave}: loop start -> wait for my own pieces (k-step 14): median {int(np.median(full[:, 6] - full[:, 0]))}, the wait itself: {int(np.median(full[:, 7] - full[:, 6]))}, "
          f"rest of the loop: {int(np.median(full[:, 1] - full[:, 7]))}")
    d = np.diff(full[:, :6], axis=1)
    for i, nm in enumerate(["MFMA loop (or: transfers issued and landed)", "wait: every wave's pieces landed", "split share", "barrier", "epilogue (waves 0..3)"]):
        print(f"   wave {stamp_wave}: {nm:46s} median {int(np.median(d[:, i])):7d}  p10 {int(np.percentile(d[:, i], 10)):7d}  p90 {int(np.percentile(d[:, i], 90)):7d}")
    sys.exit(0)
flops = 2.0 * n_img * 576 * 64 * (288 + 32)
cases = [("product", lib)] + [("alt:" + k, v) for k, v in alts.items()]
times = {n: [] for n, _ in cases}
random.seed(1)
for rnd in range(int(os.environ.get("AB_ROUNDS", "12"))):
    for name, l in random.sample(cases, len(cases)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            run(l)
        e0.record()
        for _ in range(4):
            run(l)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 4)
for name, _ in cases:
    t = times[name]
    print(f"{name:22s} median {statistics.median(t)*1e3:8.1f} us ({flops/statistics.median(t)/1e9:6.1f} TF-equivalent)   min {min(t)*1e3:8.1f} us")
