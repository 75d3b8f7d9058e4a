#!/usr/bin/env python3
"""Diagnostic (not part of the product): builds conv_igemm.hip with -DUT_STAMPS into a scratch library and
prints per-workgroup phase timings (s_memtime) of one layer-shaped convolution.
    python tools/diag/conv_stamps.py <cin> <cout> <hw> <n_img> [res]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "absolutetrack_amd", "csrc")
OUT = "/tmp/libconvdiag.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-DUT_STAMPS", "-o", OUT, os.path.join(CSRC, "conv_igemm.hip"), os.path.join(CSRC, "conv_patch.hip"),
                       os.path.join(ROOT, "tools", "diag", "conv_diag_entry.hip"), "-I", CSRC])
lib = ctypes.CDLL(OUT)
cin, cout, hw, n_img = (int(a) for a in sys.argv[1:5])
use_res = len(sys.argv) > 5
dev = "cuda:0"
x = torch.rand(n_img, hw, hw, cin, device=dev)
k_total = 9 * cin
w = torch.randn(128 * ((cout + 127) // 128), k_total, device=dev) * 0.05
bias = torch.zeros(w.shape[0], device=dev)
res = torch.rand(n_img, hw, hw, cout, device=dev) if use_res else None
out = torch.empty(n_img, hw, hw, cout, device=dev)
stamps = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
lib.conv_diag.restype = ctypes.c_int
for it in range(3):
    stamps.zero_()
    torch.cuda.synchronize()
    rc = lib.conv_diag(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                       ctypes.c_void_p(res.data_ptr() if use_res else 0), ctypes.c_void_p(out.data_ptr()),
                       n_img, hw, cin, cout, k_total, ctypes.c_void_p(stamps.data_ptr()))
    torch.cuda.synchronize()
    assert rc == 0, rc
s = stamps.cpu().numpy()
s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
print("workgroups stamped:", len(s))
names = ["start", "first_sync", "tile0_loop_end", "tile0_last_chunk_end", "tile0_epilogue_end", "tile1_epilogue_end", "end", "tiles_done"]
d = s.astype(np.float64)
print("kernel span (cycles of s_memtime @100MHz*? units):", d[:, 6].max() - t0)
for i in range(1, 7):
    v = d[:, i] - d[:, i - 1]
    print(f"{names[i-1]:>22s} -> {names[i]:<22s} median {np.median(v):10.0f}  p10 {np.percentile(v,10):10.0f}  p90 {np.percentile(v,90):10.0f}")
print("tiles per workgroup: median", np.median(d[:, 7]), "total time per wg median", np.median(d[:, 6] - d[:, 0]))
print("start spread:", np.percentile(d[:, 0] - t0, [0, 50, 100]), " end spread:", np.percentile(d[:, 6] - t0, [0, 50, 100]))
