#!/usr/bin/env python3
"""Diagnostic (not part of the product): interleaved A/B timing of conv_igemm variants on one layer shape.
Every variant is built into its own scratch library (source file + flags), all are loaded, and the launches are
timed in interleaved rounds so that clock / thermal drift hits all variants alike; reports median and min.
    python tools/diag/conv_ab.py <cin> <cout> <hw> <n_img> name=src.hip[:flag,flag,env:KEY=VAL] ...
e.g. python tools/diag/conv_ab.py 128 128 12 4096 base=absolutetrack_amd/csrc/conv_igemm.hip fine=/tmp/x.hip:-DUT_FINE_FETCH"""
import ctypes
import os
import statistics
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "absolutetrack_amd", "csrc")
cin, cout, hw, n_img = (int(a) for a in sys.argv[1:5])
dev = "cuda:0"
x = torch.rand(n_img, hw, hw, cin, device=dev)
k_total = 9 * cin
w = torch.randn(128 * ((cout + 127) // 128), k_total, device=dev) * 0.05
bias = torch.zeros(w.shape[0], device=dev)
res = torch.rand(n_img, hw, hw, cout, device=dev)
out = torch.empty(n_img, hw, hw, cout, device=dev)
flops = 2.0 * n_img * hw * hw * cout * k_total
libs = []
for spec in sys.argv[5:]:
    name, rest = spec.split("=", 1)
    src, _, fl = rest.partition(":")
    flags = [f for f in fl.split(",") if f and not f.startswith("env:")]
    envs = [f[4:].split("=", 1) for f in fl.split(",") if f.startswith("env:")]   # read by the library at its first call
    so = f"/tmp/libconvab_{name}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", *flags,
                           "-o", so, os.path.join(ROOT, src) if not os.path.isabs(src) else src,
                           os.path.join(CSRC, "conv_patch.hip"), os.path.join(ROOT, "tools", "diag", "conv_ws_experiment.hip"),
                           os.path.join(ROOT, "tools", "diag", "conv_diag_entry.hip"),
                           "-I", CSRC])
    lib = ctypes.CDLL(so)
    lib.conv_diag.restype = ctypes.c_int
    libs.append((name, lib, envs))


def run(lib):
    return lib.conv_diag(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                         ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(out.data_ptr()), n_img, hw, cin, cout, k_total, None)


ref = None
for name, lib, envs in libs:
    for k, v in envs:
        os.environ[k] = v
    for _ in range(2):
        assert run(lib) == 0
    torch.cuda.synchronize()
    for k, _v in envs:
        os.environ.pop(k, None)
    o = out.clone()
    if ref is None:
        ref = o
    else:
        print(f"{name}: max |out - {libs[0][0]}| = {float((o - ref).abs().max()):.3e}")
times = {name: [] for name, _l, _e in libs}
for rnd in range(10):
    for name, lib, _e in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            run(lib)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 4)
for name, _l, _e in libs:
    t = times[name]
    med, mn = statistics.median(t), min(t)
    print(f"{name:24s} median {med*1e3:8.1f} us ({flops/med/1e9:6.1f} TF)   min {mn*1e3:8.1f} us ({flops/mn/1e9:6.1f} TF)")
