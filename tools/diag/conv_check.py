#!/usr/bin/env python3
"""Diagnostic: one convolution through launch_conv_igemm vs torch's conv2d on the GPU.
    python tools/diag/conv_check.py <cin> <cout> <hw> <n_img> [res]"""
import ctypes, os, subprocess, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "absolutetrack_amd", "csrc")
OUT = "/tmp/libconvdiag_chk.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUT_STAMPS",
                       "-o", OUT, os.path.join(CSRC, "conv_igemm.hip"), os.path.join(CSRC, "conv_patch.hip"), os.path.join(ROOT, "tools", "diag", "conv_diag_entry.hip"), "-I", CSRC])
lib = ctypes.CDLL(OUT)
cin, cout, hw, n_img = (int(a) for a in sys.argv[1:5])
use_res = len(sys.argv) > 5
dev = "cuda:0"
torch.manual_seed(0)
x = torch.rand(n_img, hw, hw, cin, device=dev)
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
cs = 32 if cin % 32 == 0 else cin
# pack [cout_pad][k] with k = slice*(9*cs) + tap*cs + c
cout_pad = (cout + 127) // 128 * 128
wp = torch.zeros(cout_pad, 9 * cin, device=dev)
wk = wt.permute(0, 2, 3, 1).reshape(cout, 9, cin)            # [o][tap][c]
wk = wk.reshape(cout, 9, cin // cs, cs).permute(0, 2, 1, 3).reshape(cout, 9 * cin)
wp[:cout] = wk
bias = torch.zeros(cout_pad, device=dev); bias[:cout] = torch.randn(cout, device=dev)
res = torch.rand(n_img, hw, hw, cout, device=dev) if use_res else None
out = torch.full((n_img, hw, hw, cout), float("nan"), device=dev)
stamps = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
lib.conv_diag.restype = ctypes.c_int
rc = lib.conv_diag(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                   ctypes.c_void_p(res.data_ptr() if use_res else 0), ctypes.c_void_p(out.data_ptr()), n_img, hw, cin, cout, 9 * cin,
                   ctypes.c_void_p(stamps.data_ptr()))
torch.cuda.synchronize()
assert rc == 0, rc
ref = F.conv2d(x.permute(0, 3, 1, 2), wt, bias[:cout], padding=1).permute(0, 2, 3, 1)
if use_res:
    ref = ref + res
ref = torch.relu(ref)
d = (out - ref).abs()
print("max abs diff", d.max().item(), "nan count", torch.isnan(out).sum().item())
bad = (d > 1e-3).nonzero()
print("bad count", len(bad), "first bad", bad[:8].tolist())
if len(bad):
    i = bad[0].tolist(); print("got", out[tuple(i)].item(), "want", ref[tuple(i)].item())
    print("bad channels unique", bad[:, 3].unique().tolist()[:40]); print("bad x unique", bad[:, 2].unique().tolist()[:48])
