#!/usr/bin/env python3
"""Diagnostic (not part of the product): where a conv_igemm workgroup spends its cycles.  Builds a COPY of the product
kernel source with s_memtime stamps inserted (tile start, after every chunk, after the last chunk + epilogue of the
3rd and 4th tile of each workgroup), runs one layer shape and prints the distribution of chunk / tile-boundary times.
The product source has no stamp code; this script patches the copy at textual anchors.
    python tools/diag/conv_stamps.py <cin> <cout> <hw> <n_img>"""
import ctypes
import os
import statistics
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "absolutetrack_amd", "csrc")
cin, cout, hw, n_img = (int(a) for a in sys.argv[1:5])
src = open(os.path.join(CSRC, "conv_igemm.hip")).read()


def patch(s, anchor, add, after=True):
    assert s.count(anchor) == 1, (anchor, s.count(anchor))
    return s.replace(anchor, anchor + add if after else add + anchor)


src = patch(src, "namespace ut {\n", "__device__ long long* g_stamps = nullptr;\n")
src = patch(src, "  int buf = 0;\n", '''  int st_n = 0, tiles_done = 0;
#define ST() if (tid == 0 && blockIdx.x < 1024 && tiles_done >= 2 && tiles_done < 4 && st_n < 126) { \\
    __builtin_amdgcn_sched_barrier(0); g_stamps[blockIdx.x * 128 + st_n++] = (long long)__builtin_amdgcn_s_memtime(); \\
    __builtin_amdgcn_sched_barrier(0); }
''')
src = patch(src, "    UT_INIT_COMBINE();\n", "    ST();\n")
src = patch(src, "      UT_CHUNK_FINE(buf);\n", "      ST();\n")
src = patch(src, "    UT_CHUNK_FINE_LAST(buf, next);\n", "    ST(); ++tiles_done;\n")
# sub-stamps inside the last chunk: after the next tile's setup, after groups 0+1, after group 2 (+ the next tile's
# bias / residual requests), after the LDS-DMA wait, after the barrier; the stamp behind the macro closes the tail group
src = src.replace("    UT_SETUP(next);\n    UT_CHUNK_FINE_LAST(buf, next);", "    UT_SETUP(next);\n    ST();\n    UT_CHUNK_FINE_LAST(buf, next);")
i0 = src.index("#define UT_CHUNK_FINE_LAST(buf, TILE)")
i1 = src.index("  }\n", i0) + 4
src = src[:i0] + '''#define UT_CHUNK_FINE_LAST(buf, TILE)                                                                \\
  {                                                                                                  \\
    UT_READ(Y, buf, 1); UT_PIN(); UT_GROUP_FINE(X, 0, (buf) ^ 1);                                    \\
    UT_READ(X, buf, 2); UT_PIN(); UT_GROUP_FINE(Y, 1, (buf) ^ 1);                                    \\
    ST();                                                                                            \\
    UT_READ(Y, buf, 3); UT_PIN();                                                                    \\
    UT_STEP_INIT(X, x, 0, TILE) UT_STEP_INIT(X, y, 1, TILE) UT_STEP_INIT(X, z, 2, TILE) UT_STEP_INIT(X, w, 3, TILE) \\
    ST();                                                                                            \\
    UT_STAGE();                                                                                      \\
    ST();                                                                                            \\
    UT_BARRIER();                                                                                    \\
    ST();                                                                                            \\
    UT_READ(X, (buf) ^ 1, 0); UT_PIN(); UT_TAIL_EPI(Y); UT_PIN();                                    \\
  }
''' + src[i1:]
N_SUB = 5
var = "/tmp/conv_igemm_stamped.hip"
open(var, "w").write(src)
entry = "/tmp/conv_stamp_entry.hip"
open(entry, "w").write('''#include "ut_kernels.h"
namespace ut { extern __device__ long long* g_stamps; }
extern "C" int conv_diag(const float* in, const float* w, const float* bias, const float* res, float* out, int n_img,
                         int hw, int cin, int cout, int k_total, long long* stamps) {
  ut::ConvLaunch c{};
  c.in = in; c.w = w; c.bias = bias; c.res = res; c.out = out;
  c.n_img = n_img; c.H = hw; c.W = hw; c.cin = cin; c.Ho = hw; c.Wo = hw;
  c.cout_store = cout; c.cout_pad = (cout + 127) / 128 * 128; c.k_total = k_total; c.k_pad = k_total;
  c.cslice = cin % 32 == 0 ? 32 : cin; c.ksize = 3; c.stride = 1; c.pad = 1; c.relu = 1; c.out_nchw = 0;
  c.num_cu = 256; c.device = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(ut::g_stamps), &stamps, sizeof(stamps));
  static unsigned* cnt = nullptr;
  if (!cnt) (void)hipMalloc((void**)&cnt, 4);
  (void)hipMemsetAsync(cnt, 0, 4, 0);
  c.tile_counter = cnt;
  return (int)ut::launch_conv_igemm(c, 0);
}
''')
so = "/tmp/libconvstamps.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-fgpu-rdc",
                       "-o", so, var, os.path.join(CSRC, "conv_patch.hip"), entry, "-I", CSRC])
lib = ctypes.CDLL(so)
dev = "cuda:0"
x = torch.rand(n_img, hw, hw, cin, device=dev)
k_total = 9 * cin
w = torch.randn(128 * ((cout + 127) // 128), k_total, device=dev) * 0.05
bias = torch.zeros(w.shape[0], device=dev)
res = torch.rand(n_img, hw, hw, cout, device=dev)
out = torch.empty(n_img, hw, hw, cout, device=dev)
stamps = torch.zeros(1024, 128, dtype=torch.int64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(3):
    stamps.zero_()
    assert lib.conv_diag(p(x), p(w), p(bias), p(res), p(out), n_img, hw, cin, cout, k_total, p(stamps)) == 0
    torch.cuda.synchronize()
st = stamps.cpu().numpy()
n_chunks = k_total // 32
per_tile = n_chunks + 1 + N_SUB   # tile start, after each steady chunk (n_chunks - 1), [sub-stamps], after the last chunk
rows = st[(st != 0).sum(1) >= 2 * per_tile]
print(f"{rows.shape[0]} workgroups with two stamped tiles; {n_chunks} chunks per tile")
t = rows[:, : 2 * per_tile].reshape(-1, 2, per_tile)
chunk = np.diff(t[:, :, : n_chunks], axis=2).reshape(-1)            # steady chunks
last = (t[:, :, -1] - t[:, :, n_chunks - 1]).reshape(-1)             # setup + last chunk + epilogue
gap = (t[:, 1, 0] - t[:, 0, -1])                                     # end of tile -> start of next (combine etc.)
tile = (t[:, 1, 0] - t[:, 0, 0])
q = lambda a: " ".join(f"{np.percentile(a, pc):8.0f}" for pc in (5, 25, 50, 75, 95))
print("percentiles (cycles)              5%      25%      50%      75%      95%     mean")
print(f"steady chunk                {q(chunk)} {chunk.mean():8.0f}")
print(f"setup + last chunk + epilogue {q(last)} {last.mean():8.0f}")
names = ["setup of the next tile", "groups 0+1 (+8 pieces)", "group 2 (+ bias/res requests)", "LDS-DMA wait", "barrier", "tail group + stores"]
for k, nm in enumerate(names):
    d = (t[:, :, n_chunks + k] - t[:, :, n_chunks + k - 1]).reshape(-1)
    print(f"  {nm:30s}{q(d)} {d.mean():8.0f}")
print(f"tile end -> next start      {q(gap)} {gap.mean():8.0f}")
print(f"whole tile                  {q(tile)} {tile.mean():8.0f}")
by_pos = np.diff(t[:, :, : n_chunks], axis=2).reshape(-1, n_chunks - 1).mean(0)
print("mean steady-chunk time by chunk position:", " ".join(f"{v:.0f}" for v in by_pos))
