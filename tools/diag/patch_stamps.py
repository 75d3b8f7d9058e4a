#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime stamps of the second tile of every workgroup of conv3x3_c32_patch_kernel."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "absolutetrack_amd", "csrc")
OUT = "/tmp/libconvdiag_patch.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUT_STAMPS", "-o", OUT,
                       os.path.join(CSRC, "conv_igemm.hip"), os.path.join(CSRC, "conv_patch.hip"),
                       os.path.join(ROOT, "tools", "diag", "conv_diag_entry.hip"), "-I", CSRC])
lib = ctypes.CDLL(OUT); lib.conv_diag.restype = ctypes.c_int
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda:0"
x = torch.rand(n_img, 48, 48, 32, device=dev); w = torch.randn(128, 288, device=dev) * 0.05; b = torch.zeros(128, device=dev)
res = torch.rand(n_img, 48, 48, 32, device=dev); out = torch.empty_like(res)
st = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
for _ in range(3):
    st.zero_(); torch.cuda.synchronize()
    rc = lib.conv_diag(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(res.data_ptr()),
                       ctypes.c_void_p(out.data_ptr()), n_img, 48, 32, 32, 288, ctypes.c_void_p(st.data_ptr()))
    torch.cuda.synchronize(); assert rc == 0
s = st.cpu().numpy().astype(np.float64); s = s[s[:, 0] != 0]
names = ["tile_start", "after_issue(next patch, init loads)", "after_mfma", "after_wait+ready", "after_stores", "after_barrier"]
print("workgroups", len(s))
for i in range(1, 6):
    v = s[:, i] - s[:, i - 1]
    print(f"{names[i-1]:>40s} -> {names[i]:<40s} median {np.median(v):9.0f} p10 {np.percentile(v,10):9.0f} p90 {np.percentile(v,90):9.0f}")
print("tile total median", np.median(s[:, 5] - s[:, 0]))
