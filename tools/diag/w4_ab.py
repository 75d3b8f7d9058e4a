"""A/B timing, bit checks, timing-only ablations and phase stamps of csrc/conv_w4.hip against the chunked kernel
(conv_split_kernel<256, 128, 4, 2, true>) on one convolution.  The product source carries no timing code: every variant is a text patch
of a COPY.
    python tools/diag/w4_ab.py <cin> <cout> <hw> <n_img>
      W4_ABL=nostore,nores,nowload,nopatch,noxread,nobarrier   timing-only ablations (wrong numbers)
      W4_ALT_SRCS=<path>,<path>                               other versions of conv_w4.hip (checked bit for bit, timed as alt:<file>)
      W4_STAMPS=1                                             s_memtime stamps of every workgroup's fourth tile (wave 0)
      AB_ROUNDS=<n>                                           timing rounds (default 10; random order, a lead-in per case)
      W4_STRIDE=2                                             the phase-plane form (hw = the INPUT map; no residual; against the chunked
                                                              gather kernel and the fp32-instruction kernel: sums in another order)"""
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
cin, cout, hw, n_img = (int(a) for a in sys.argv[1:5])
stride = int(os.environ.get("W4_STRIDE", "1"))
ho = hw // stride

ABL = {
    "nostore": [("            __builtin_amdgcn_raw_buffer_store_b32(o, o_rsrc, off, 0, 0);", "            if (p.k_pad < 0) __builtin_amdgcn_raw_buffer_store_b32(o, o_rsrc, off, 0, 0);")],
    "nores": [("__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off, 0, 0)", "__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, W4_HOOB, 0, 0)")],
    "nowload": [("    wf[SET][PL] = __builtin_bit_cast(", "    if (p.k_pad < 0) wf[SET][PL] = __builtin_bit_cast(")],
    "nopatch": [("          if ((N) == 14 && q < W4_NLOAD)  ", "          if ((N) == 14 && q < W4_NLOAD && p.k_pad < 0)  "),
                ("          if ((N) == 13 && q >= 4 && q < 4 + W4_NLOAD) {", "          if ((N) == 13 && q >= 4 && q < 4 + W4_NLOAD && p.k_pad < 0) {")],
    "noxread": [("  xp[I][PC] = *reinterpret_cast<const u32x4w*>(smem + (BASE)[I]", "  if (p.k_pad < 0) xp[I][PC] = *reinterpret_cast<const u32x4w*>(smem + (BASE)[I]")],
    "s2noload": [("          if ((N) >= 22 && (N) <= 24 && (N) - 22 < W4S_LD_N(q))  ", "          if ((N) >= 22 && (N) <= 24 && (N) - 22 < W4S_LD_N(q) && p.k_pad < 0)  ")],
    "s2nocv": [("          if ((N) >= 10 && (N) <= 21 && ((N) - 10) / 4 < W4S_CV_N(q)) {  ", "          if ((N) >= 10 && (N) <= 21 && ((N) - 10) / 4 < W4S_CV_N(q) && p.k_pad < 0) {  ")],
    "s2nobarrier": [("            __builtin_amdgcn_s_barrier();                                                            \\\n", "            if (p.k_pad < 0) __builtin_amdgcn_s_barrier();                                                            \\\n")],
    "nobarrier": [("          __builtin_amdgcn_s_barrier();                                                              \\\n", "          if (p.k_pad < 0) __builtin_amdgcn_s_barrier();                                                              \\\n")],
}
STAMP = ('if (n_done == 3) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(st_[I]) :: "memory"); '
         '__builtin_amdgcn_sched_barrier(0); }')
st = lambda i: STAMP.replace("[I]", f"[{i}]")
STAMP_PATCHES = [
    ("    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);\n", ""),      # p.status is the stamp buffer here
    ("  for (;;) {\n    // the tile after this one", "  unsigned long long st_[8];\n  int n_done = 0;\n  for (;;) {\n    " + st(0) + "\n    // the tile after this one"),
    ("        const int ws = q % W4_NSET;                                                                  \\\n",
     "        const int ws = q % W4_NSET;                                                                  \\\n"
     "        if (q == 9 && sl == 0) " + st(1) + " if (q == 9 && last_slice) " + st(5) + " \\\n"),
    ("      cur_buf = b1;\n", "      cur_buf = b1;\n      if (sl == 0) " + st(2) + "\n      if (sl == 1) " + st(3) + "\n      if (sl == n_slices - 2) " + st(4) + "\n      if (sl == n_slices - 1) " + st(6) + "\n"),
    ("    if ((unsigned)next_tile >= (unsigned)n_tiles) break;\n",
     "    " + st(7) + "\n    if (n_done == 3 && tid == 0 && p.status) {\n      unsigned long long* d = reinterpret_cast<unsigned long long*>(p.status) + blockIdx.x * 8;\n"
     "      for (int i = 0; i < 8; ++i) d[i] = st_[i];\n    }\n    ++n_done;\n    if ((unsigned)next_tile >= (unsigned)n_tiles) break;\n"),
]

# the phase-plane form: stamps behind slices 0, 1, n - 2, n - 1 and behind the epilogue
STAMP_PATCHES_S2 = [
    STAMP_PATCHES[0],
    ("  for (;;) {\n    // the tile after this one", "  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n  int n_done = 0;\n  for (;;) {\n    " + st(0) + "\n    // the tile after this one"),
    ("        W4S_SLICE()\n", "        W4S_SLICE()\n        if (sl == 0) " + st(1) + "\n        if (sl == 1) " + st(2) + "\n        if (sl == n_slices - 2) " + st(3) + "\n        if (sl == n_slices - 1) " + st(4) + "\n"),
    (STAMP_PATCHES[4][0], STAMP_PATCHES[4][1].replace(st(7), st(5))),
]


def patched(name, patches, src=None):
    text = open(src or os.path.join(CSRC, "conv_w4.hip")).read()
    for old, new in patches:
        assert old in text, old
        text = text.replace(old, new)
    path = f"/tmp/conv_w4_{name}.hip"
    open(path, "w").write(text)
    return path


def build(name, w4=None, flags=()):
    so = f"/tmp/libw4ab_{name}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", *flags, "-o", so,
                           os.path.join(CSRC, "conv_igemm.hip"), os.path.join(CSRC, "conv_patch.hip"), os.path.join(ROOT, "tools", "diag", "conv_c64r.hip"),
                           os.path.join(CSRC, "conv_c64k.hip"), w4 or os.path.join(CSRC, "conv_w4.hip"), os.path.join(CSRC, "conv_split.hip"),
                           os.path.join(ROOT, "tools", "diag", "split_entry.hip"), "-I", CSRC])
    return ctypes.CDLL(so)


stamps = bool(os.environ.get("W4_STAMPS"))
stamp_src = os.environ.get("W4_STAMPS") if os.path.exists(os.environ.get("W4_STAMPS", "")) else None      # W4_STAMPS=<path>: stamps of that source
lib = build("stamps", patched("stamps", STAMP_PATCHES if stride == 1 else STAMP_PATCHES_S2, stamp_src), flags=["-DC64_STAMPS"]) if stamps else build("product")
vlibs = {}
for nm in [q for q in os.environ.get("W4_ABL", "").split(",") if q]:
    vlibs["abl:" + nm] = build("abl_" + nm.replace("+", "_"), patched(nm.replace("+", "_"), sum((ABL[x] for x in nm.split("+")), [])))
alts = {}
for path in [q for q in os.environ.get("W4_ALT_SRCS", "").split(",") if q]:
    nm = os.path.splitext(os.path.basename(path))[0]
    alts["alt:" + nm] = build("alt_" + nm, path)
if "--build-only" in sys.argv:
    sys.exit(0)

dev = "cuda:0"
torch.manual_seed(0)
x = torch.rand(n_img, hw, hw, cin, device=dev) * 2 - 0.5
k_total = 9 * cin
cout_pad = 128 * ((cout + 127) // 128)
w_oihw = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cout)) ** 0.5
wp = torch.zeros(cout_pad, k_total)
wp[:cout] = w_oihw.reshape(cout, cin // 32, 32, 9).permute(0, 1, 3, 2).reshape(cout, k_total)
bias = torch.zeros(cout_pad)
bias[:cout] = torch.randn(cout) * 0.1
res = torch.rand(n_img, ho, ho, cout, device=dev) if stride == 1 else None
split = np.zeros(2 * cout_pad * k_total, np.uint16)
assert lib.split_pack(wp.numpy().ctypes.data_as(ctypes.c_void_p), cout_pad, k_total, split.ctypes.data_as(ctypes.c_void_p)) == 0
for l in list(vlibs.values()) + list(alts.values()):
    assert l.split_pack(wp.numpy().ctypes.data_as(ctypes.c_void_p), cout_pad, k_total, split.ctypes.data_as(ctypes.c_void_p)) == 0
w_d, b_d = wp.to(dev), bias.to(dev)
s_d = torch.from_numpy(split.view(np.int16)).to(dev)
out = torch.empty(n_img, ho, ho, cout, device=dev)


def run(mode, l=lib):
    rc = l.conv_diag2(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w_d.data_ptr()), ctypes.c_void_p(s_d.data_ptr()),
                      ctypes.c_void_p(b_d.data_ptr()), ctypes.c_void_p(res.data_ptr() if res is not None else None), ctypes.c_void_p(out.data_ptr()),
                      n_img, hw, cin, cout, 3, stride, 1, mode)
    assert rc == 0, rc


if stamps:
    run(1)
    buf = np.zeros(256 * 8, np.uint64)
    assert lib.conv_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(256, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    if stride != 1:
        n_sl = cin // 16
        names = ["slice 0", "slice 1", f"slices 2 .. {n_sl - 2} ({n_sl - 3} of them)", f"slice {n_sl - 1}", "epilogue"]
        d = np.diff(t[:, :6], axis=1)
        for i, nm in enumerate(names):
            print(f"   {nm:40s} median {int(np.median(d[:, i])):7d}  p10 {int(np.percentile(d[:, i], 10)):7d}  p90 {int(np.percentile(d[:, i], 90)):7d}")
        print(f"   whole tile median {int(np.median(t[:, 5] - t[:, 0]))} (s_memtime ticks; a slice is 243 MFMAs per wave = 7,776 cycles of the matrix pipe)")
        sys.exit(0)
    d = np.diff(t, axis=1)
    names = ["first slice, k-steps 0 .. 8", "first slice, k-steps 9 .. 17", "second slice", "slices 2 .. n-2", "last slice, k-steps 0 .. 8 (fetches the next tile's patch)",
             "last slice, k-steps 9 .. 17", "epilogue"]
    for i, nm in enumerate(names):
        print(f"   {nm:62s} median {int(np.median(d[:, i])):7d}  p10 {int(np.percentile(d[:, i], 10)):7d}  p90 {int(np.percentile(d[:, i], 90)):7d}")
    print(f"   whole tile median {int(np.median(t[:, 7] - t[:, 0]))} (s_memtime ticks; a slice is 432 MFMAs per wave = 13,824 cycles of the matrix pipe)")
    sys.exit(0)

out.fill_(float("nan"))
run(3)
torch.cuda.synchronize()
ref = out.clone()
out.fill_(float("nan"))
run(1)
torch.cuda.synchronize()
print(f"conv_w4 == chunked bit for bit: {bool(torch.equal(out, ref))}   max |difference| {float((out - ref).abs().max()):.3e} (|out| max {float(ref.abs().max()):.2f})")
if stride != 1:
    got = out.clone()
    out.fill_(float("nan"))
    run(0)
    torch.cuda.synchronize()
    print(f"   against the fp32-instruction kernel: conv_w4 {float((got - out).abs().max()):.3e}, chunked {float((ref - out).abs().max()):.3e}; "
          f"against torch conv2d (fp64): {float((got.double() - torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w_oihw.to(dev).double(), bias[:cout].to(dev).double(), stride=stride, padding=1).relu().permute(0, 2, 3, 1)).abs().max()):.3e}")
for nm, l in alts.items():
    out.fill_(float("nan"))
    run(1, l)
    torch.cuda.synchronize()
    print(f"{nm} == chunked bit for bit: {bool(torch.equal(out, ref))}")
flops = 2.0 * n_img * ho * ho * cout * k_total
cases = [("conv_w4", 1, lib), ("chunked HALO", 3, lib)] + ([("conv_c64k", 4, lib)] if cout == 64 else []) + [(v, 1, l) for v, l in vlibs.items()] + [(v, 1, l) for v, l in alts.items()]
times = {name: [] for name, _m, _l in cases}
random.seed(1)
for rnd in range(int(os.environ.get("AB_ROUNDS", "10"))):
    for name, mode, l in random.sample(cases, len(cases)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            run(mode, l)
        e0.record()
        for _ in range(4):
            run(mode, l)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 4)
for name, _m, _l in cases:
    t = times[name]
    med, mn = statistics.median(t), min(t)
    print(f"{name:24s} median {med*1e3:8.1f} us ({flops/med/1e9:6.1f} TF-equivalent)   min {mn*1e3:8.1f} us ({flops/mn/1e9:6.1f})")
