#!/usr/bin/env python3
"""Diagnostic (not part of the product): times one layer-shaped convolution built with the timing-only
ablation flags of conv_igemm.hip (results are wrong in ablated builds; only the duration matters).
    python tools/diag/conv_ablate.py <cin> <cout> <hw> <n_img>"""
import ctypes
import os
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
variants = [("baseline", []), ("no_A_loads", ["-DUT_DIAG_NO_A"]), ("no_B_loads", ["-DUT_DIAG_NO_B"]),
            ("no_A_no_B_loads(addr math only)", ["-DUT_DIAG_NO_A", "-DUT_DIAG_NO_B"]), ("no_fetch", ["-DUT_DIAG_NO_FETCH"]), ("no_fetch+no_stage", ["-DUT_DIAG_NO_FETCH", "-DUT_DIAG_NO_STAGE"]),
            ("no_barrier", ["-DUT_DIAG_NO_BARRIER"]),
            ("no_fetch+no_stage+no_barrier", ["-DUT_DIAG_NO_FETCH", "-DUT_DIAG_NO_STAGE", "-DUT_DIAG_NO_BARRIER"])]
only = os.environ.get("UT_ABLATE_ONLY")
if only:
    variants = [v for v in variants if v[0] in only.split(",")]
extra = os.environ.get("UT_ABLATE_FLAGS", "").split()
for name, flags in variants:
    so = "/tmp/libconvdiag_" + "".join(ch if ch.isalnum() else "_" for ch in name) + ".so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUT_STAMPS",
                           *flags, *extra, "-o", so, os.path.join(CSRC, "conv_igemm.hip"), os.path.join(CSRC, "conv_patch.hip"),
                           os.path.join(ROOT, "tools", "diag", "conv_diag_entry.hip"), "-I", CSRC])
    lib = ctypes.CDLL(so)
    lib.conv_diag.restype = ctypes.c_int
    stamps = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
    def run():
        return lib.conv_diag(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                             ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(out.data_ptr()), n_img, hw, cin, cout, k_total,
                             ctypes.c_void_p(stamps.data_ptr()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:32s} {ms*1e3:9.1f} us   {flops/ms/1e9:7.1f} TFLOP/s")
