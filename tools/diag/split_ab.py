#!/usr/bin/env python3
"""Diagnostic (not part of the product): the split-fp16 convolution (conv_split.hip) against the fp32-MFMA one
(conv_igemm.hip) on one layer shape - error of both against a float64 convolution of the first images, and interleaved timing.
    python tools/diag/split_ab.py <cin> <cout> <hw> <n_img> [ksize] [stride]"""
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
ksize = int(sys.argv[5]) if len(sys.argv) > 5 else 3
stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
# variants of conv_split.hip (text patches of a copy, see VARIANTS) are timed beside the product kernel: name,name,...
VARIANTS = {
    # no fp32 -> fp16 conversion work (wrong numbers): what the VALU split costs
    "noconv": [("  const f16x2 h = ", "  p0 = __float_as_uint(a); p1 = __float_as_uint(b); return;\n  const f16x2 h = ")],
    # no transfers after the prologue (stale operands): what the fetch costs
    "nodma": [("#define SP_A_ISSUE(I, OFF) dma_piece(", "#define SP_A_ISSUE(I, OFF) if (p.k_pad < 0) dma_piece("),
              ("#define SP_H_ISSUE(Q, SLICE) dma_piece(", "#define SP_H_ISSUE(Q, SLICE) if (p.k_pad < 0) dma_piece("),
              ("    dma_piece(w_words, smem_addr + W_BASE + fw_stage", "    if (p.k_pad < 0) dma_piece(w_words, smem_addr + W_BASE + fw_stage")],
    # no chunk synchronisation (racy: timing only)
    "nosync": [("    asm volatile(\"s_mov_b64 %0, exec\\n\\ts_mov_b64 exec, 1\\n\\tds_add_u32 %1, %2", "    if (p.k_pad < 0) asm volatile(\"s_mov_b64 %0, exec\\n\\ts_mov_b64 exec, 1\\n\\tds_add_u32 %1, %2"),
               ("  if ((int)(__builtin_amdgcn_readfirstlane(PEEKED) - (TARGET)) < 0) {", "  if (p.k_pad < 0) {"),
               ("#define SP_PEEK(ADDR) (*reinterpret_cast<volatile lds_u32*>(ADDR))", "#define SP_PEEK(ADDR) 0u")],
    # READ signal without its lgkmcnt(0) (racy: timing only)
    "nolgkm": [("SP_SIGNAL(read_addr, 1);", "SP_SIGNAL(read_addr, 0);")],
    # signals and looks, but no waiting on the counters (racy: timing only)
    "noawait": [("  if ((int)(__builtin_amdgcn_readfirstlane(PEEKED) - (TARGET)) < 0) {", "  if (p.k_pad < 0 && (int)(__builtin_amdgcn_readfirstlane(PEEKED) - (TARGET)) < 0) {")],
    # no counted vmcnt wait at the arrival (racy: timing only)
    "nowait": [("      if (skip_waits > 0) --skip_waits;  ", "      if (p.k_pad >= 0) {} else if (skip_waits > 0) --skip_waits;  ")],
}
# no output stores / no residual requests (wrong numbers): what the epilogue's memory operations cost
VARIANTS["nostore"] = [("            __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, off, 0, 0);", "            if (p.k_pad < 0) __builtin_amdgcn_raw_buffer_store_b128(pk, o_rsrc, off, 0, 0);")]
VARIANTS["nores"] = [("          rr[i][j][g4] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0);", "          rr[i][j][g4] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, OOB, 0, 0);")]
VARIANTS["noearlywait"] = [("      if constexpr (EARLY_RES) asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");", "")]
# text patches of a COPY of conv_c64k.hip (the product source carries no timing code): C64_STAMPS=1|2 phase stamps of the fourth tile
# of every workgroup (wave 0 / wave 4 of the workgroup writes), C64K_ABL=noconvert|noreads timing ablations (wrong numbers)
C64K_STAMP = ('if (k == 3) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(st_[I]) :: "memory"); '
              '__builtin_amdgcn_sched_barrier(0); }')
def c64k_stamp_patches(which):
    st = lambda i: C64K_STAMP.replace("[I]", f"[{i}]")
    return [
        ("    if (!ok && tid == 0 && blockIdx.x == 0 && p.status) atomicOr(p.status, UT_SPLIT_RANGE);\n", ""),      # p.status is the stamp buffer here
        ("  for (int k = 0; k < my_tiles; ++k) {\n", "  unsigned long long st_[8];\n  for (int k = 0; k < my_tiles; ++k) {\n"),
        ("    {\n      u32x4k pxA[2][2], pxB[2][2];", "    " + st(0) + "\n    {\n      u32x4k pxA[2][2], pxB[2][2];"),
        ("    K_DRAIN();\n", "    K_DRAIN();\n    " + st(1) + "\n    if (k == 3) { st_[2] = st_[1]; st_[3] = st_[1]; }\n"),
        ("      K_AWAIT(cnt_addr + 4u * (unsigned)ks, 4 * (k + 1))\n", "      " + st(2) + "\n      K_AWAIT(cnt_addr + 4u * (unsigned)ks, 4 * (k + 1))\n      " + st(3) + "\n"),
        ("    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    __builtin_amdgcn_s_barrier();\n\n    if (ks == 0) {",
         "    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    " + st(4) + "\n    __builtin_amdgcn_s_barrier();\n\n    if (ks == 0) {"),
        ("      K_SIGNAL(cnt_addr + 8u);\n    }\n",
         "      K_SIGNAL(cnt_addr + 8u);\n    }\n    " + st(5) + "\n    if (k == 3 && tid == %d && p.status) {\n      unsigned long long* d = reinterpret_cast<unsigned long long*>(p.status) + blockIdx.x * 8;\n"
         "      for (int i = 0; i < 6; ++i) d[i] = st_[i];\n    }\n" % (256 * (which - 1))),
    ]
C64K_ABL = {
    "noconvert": [("      convert_own_slice((k + 1) & 1);\n", "")],
    "noreads": [("#define K_LOAD(DST, ADDR, S)                                                                         \\\n  {", "#define K_LOAD(DST, ADDR, S)                                                                         \\\n  if (p.k_pad < 0) {")],
}
def patched_c64k(name, patches):
    text = open(os.path.join(CSRC, "conv_c64k.hip")).read()
    for old, new in patches:
        assert old in text, old
        text = text.replace(old, new, 1)
    path = f"/tmp/conv_c64k_{name}.hip"
    open(path, "w").write(text)
    return path
variants = [v for v in os.environ.get("SPLIT_VARIANTS", "").split(",") if v]
# SPLIT_ALT_SRC=<path>: another conv_split.hip (e.g. an earlier commit's, `git show <rev>:absolutetrack_amd/csrc/conv_split.hip`) as variant "alt"
ALT_SRC = os.environ.get("SPLIT_ALT_SRC")


def build(name, patches, flags=(), alt=None, c64=None, c64k=None, w4=None):
    src = alt or os.path.join(CSRC, "conv_split.hip")
    if patches:
        text = open(src).read()
        for old, new in patches:
            assert old in text, old
            text = text.replace(old, new)
        src = f"/tmp/conv_split_{name}.hip"
        open(src, "w").write(text)
    c64_src = c64 or os.path.join(ROOT, "tools", "diag", "conv_c64r.hip")
    so = f"/tmp/libsplitab_{name}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", *flags, "-o", so,
                           os.path.join(CSRC, "conv_igemm.hip"), os.path.join(CSRC, "conv_patch.hip"), c64_src, c64k or os.path.join(CSRC, "conv_c64k.hip"), w4 or os.path.join(CSRC, "conv_w4.hip"), src,
                           os.path.join(ROOT, "tools", "diag", "split_entry.hip"), "-I", CSRC])
    return ctypes.CDLL(so)


if os.environ.get("C64_STAMPS"):      # phase stamps of conv_c64k.hip (fourth tile of every workgroup; 1: the epilogue wave of a pair, 2: the other)
    lib = build("stamps", [], flags=["-DC64_STAMPS"], c64k=patched_c64k("stamps", c64k_stamp_patches(int(os.environ["C64_STAMPS"]))))
else:
    lib = build("product", [], c64k=os.environ.get("C64K_MAIN_SRC"))      # C64K_MAIN_SRC=<path>: another conv_c64k.hip as the main build
vlibs = {}
for v in variants:
    patches, flags = [], []
    for part in v.split("+"):
        if part.startswith("-D"):
            flags.append(part)
        else:
            patches += VARIANTS[part]
    vlibs[v] = build(v.replace("=", "_"), patches, flags)
if ALT_SRC:
    vlibs["alt"] = build("alt", [], alt=ALT_SRC)
# C64_ALT_SRCS=<path>,<path>: other versions of conv_c64r.hip (with the product conv_split.hip), timed beside the product as c64:<file>
for nm in [q for q in os.environ.get("C64K_ABL", "").split(",") if q]:      # timing ablations of conv_c64k.hip (wrong numbers)
    vlibs["c64k:" + nm] = build("c64k_" + nm, [], c64k=patched_c64k(nm, C64K_ABL[nm]))
for path in [q for q in os.environ.get("C64K_ALT_SRCS", "").split(",") if q]:      # likewise for conv_c64k.hip
    nm = os.path.splitext(os.path.basename(path))[0]
    vlibs["c64k:" + nm] = build("c64k_" + nm, [], c64k=path)
for path in [q for q in os.environ.get("W4_ALT_SRCS", "").split(",") if q]:      # other versions of conv_w4.hip, timed as w4:<file>
    nm = os.path.splitext(os.path.basename(path))[0]
    vlibs["w4:" + nm] = build("w4_" + nm, [], w4=path)
for path in [q for q in os.environ.get("C64_ALT_SRCS", "").split(",") if q]:
    nm = os.path.splitext(os.path.basename(path))[0]
    vlibs["c64:" + nm] = build("c64_" + nm, [], c64=path)
dev = "cuda:0"
torch.manual_seed(0)
ho = (hw + 2 * (ksize // 2) - ksize) // stride + 1
x = torch.rand(n_img, hw, hw, cin, device=dev) * 2 - 0.5
k_total = ksize * ksize * cin
cout_pad = 128 * ((cout + 127) // 128)
w_oihw = torch.randn(cout, cin, ksize, ksize) * (2.0 / (ksize * ksize * cout)) ** 0.5
# packed k order (channel slice of 32, tap, channel in slice)
wp = torch.zeros(cout_pad, k_total)
wp[:cout] = w_oihw.reshape(cout, cin // 32, 32, ksize * ksize).permute(0, 1, 3, 2).reshape(cout, k_total)
bias = torch.zeros(cout_pad)
bias[:cout] = torch.randn(cout) * 0.1
res = torch.rand(n_img, ho, ho, cout, device=dev)
split = np.zeros(2 * cout_pad * k_total, np.uint16)
assert lib.split_pack(wp.numpy().ctypes.data_as(ctypes.c_void_p), cout_pad, k_total, split.ctypes.data_as(ctypes.c_void_p)) == 0
w_d, b_d = wp.to(dev), bias.to(dev)
s_d = torch.from_numpy(split.view(np.int16)).to(dev)
out = torch.empty(n_img, ho, ho, cout, device=dev)


def run(mode, lib=lib):
    rc = lib.conv_diag2(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w_d.data_ptr()), ctypes.c_void_p(s_d.data_ptr()),
                        ctypes.c_void_p(b_d.data_ptr()), ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                        n_img, hw, cin, cout, ksize, stride, 1, mode)
    assert rc == 0, rc


if os.environ.get("C64_STAMPS"):
    run(1)
    buf = np.zeros(256 * 8, np.uint64)
    assert lib.conv_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(256, 8)[:, :6].astype(np.int64)
    d = np.diff(st, axis=1)
    for i, nm in enumerate(["MFMA loop (108 MFMAs)", "partial sums out (ks = 1)", "wait: my slice of the next patches landed", "split of my slice (+ residual requests: ks = 1)", "barrier (+ epilogue: ks = 0)"]):
        print(f"   {nm:48s} median {int(np.median(d[:, i])):7d}  p10 {int(np.percentile(d[:, i], 10)):7d}  p90 {int(np.percentile(d[:, i], 90)):7d}")
    sys.exit(0)
nref = min(n_img, 6)
ref = torch.nn.functional.conv2d(x[:nref].permute(0, 3, 1, 2).double().cpu(), w_oihw.double(), bias[:cout].double(), stride, ksize // 2)
ref = torch.relu(ref.permute(0, 2, 3, 1) + res[:nref].double().cpu())
outs = {}
for mode, name in ((0, "fp32 mfma"), (1, "split f16x3")):
    out.fill_(float("nan"))
    run(mode)
    torch.cuda.synchronize()
    o = out.clone()
    outs[name] = o
    assert os.environ.get("ERRMAP") or torch.isfinite(o).all(), name
    print(f"{name:14s} max |out - f64 conv| over {nref} images = {float((o[:nref].double().cpu() - ref).abs().max()):.3e}")
a, b = outs["fp32 mfma"], outs["split f16x3"]
print(f"max |split - fp32| over all {n_img} images = {float((a - b).abs().max()):.3e}   (|out| max {float(a.abs().max()):.2f})")
if cin == 64 and cout == 64 and ksize == 3 and stride == 1:
    for mode, name in ((2, "c64r"), (3, "chunked")):
        out.fill_(float("nan"))
        run(mode)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), name
        print(f"max |{name} - c64k| over all {n_img} images = {float((out - b).abs().max()):.3e}   max |{name} - f64| = {float((out[:nref].double().cpu() - ref).abs().max()):.3e}")
flops = 2.0 * n_img * ho * ho * cout * k_total
cases = [("fp32 mfma", 0, lib), ("split f16x3", 1, lib)] + [(v, 1, l) for v, l in vlibs.items()]
if cin == 64 and cout == 64 and ksize == 3 and stride == 1:
    cases += [("c64r (one wave/SIMD)", 2, lib), ("chunked 256x64 HALO", 3, lib)]
if cin >= 64 and cout % 128 == 0 and ksize == 3 and stride == 1:
    out.fill_(float("nan"))
    run(3)
    torch.cuda.synchronize()
    print(f"conv_w4 == chunked 256x128 HALO bit for bit: {bool(torch.equal(out, b))}")
    cases += [("chunked 256x128 HALO", 3, lib)]
times = {name: [] for name, _m, _l in cases}
import random
random.seed(1)
for rnd in range(int(os.environ.get("AB_ROUNDS", "10"))):
    for name, mode, l in random.sample(cases, len(cases)):      # a new order every round: position effects average out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):      # its own lead-in: the clock a case sees depends on what ran just before it (a case behind the slow fp32
            run(mode, l)        # kernel measured 10 % faster than the same code further down the list)
        e0.record()
        for _ in range(4):
            run(mode, l)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 4)
for name, _m, _l in cases:
    t = times[name]
    med, mn = statistics.median(t), min(t)
    print(f"{name:22s} median {med*1e3:8.1f} us ({flops/med/1e9:6.1f} TF-equivalent)   min {mn*1e3:8.1f} us ({flops/mn/1e9:6.1f})")
