#!/usr/bin/env python3
"""Per-layer view of a rocprofv3 --kernel-trace CSV of `bench.py` (known-skeleton mode):
maps the convolution launches (conv_igemm / conv_split / halo patch) of the LAST step to the network's convolutions by launch order and
prints duration, FLOPs and TFLOP/s of each - and the launch's COMPULSORY bytes (input once, output once, the residual where there is one;
fp32 tensors) over its duration: the HBM side of the same launch -, plus totals of every other kernel in that step.
    python tools/layer_profile.py <kernel_trace.csv> <n_crops> <chunk>"""
import collections
import csv
import sys

path, n_crops, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3

blocks = [(32, 32, 1), (32, 32, 1), (32, 64, 2), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 128, 1),
          (128, 128, 1), (128, 128, 1), (128, 256, 2), (256, 256, 1)]
def block_convs(bi, hw):
    ci, co, s = blocks[bi]
    ho = hw // s
    x, y = 4 * ci * hw * hw, 4 * co * ho * ho            # bytes of the block's input and of one of its output-sized tensors
    out = [(f"b{bi}.conv1 {ci}->{co} s{s} @{ho}", 2 * 9 * ci * co * ho * ho, x + y)]
    if s != 1 or ci != co:
        out.append((f"b{bi}.ds {ci}->{co} @{ho}", 2 * ci * co * ho * ho, 4 * ci * ho * ho + y))      # (the sampled pixels only)
    out.append((f"b{bi}.conv2 {co}->{co} @{ho}", 2 * 9 * co * co * ho * ho, 3 * y))                  # input, residual, output
    return out, ho

fused_l1 = any("conv_block32" in r["Kernel_Name"] for r in rows)   # split-fp16 mode: layer1's blocks are one launch each
fused_s2 = any("conv_c32s2" in r["Kernel_Name"] for r in rows)     # ... and layer2's stride-2 3x3 + its 1x1 shortcut
seq = []   # (name, flops, compulsory bytes) of the launch
n_chunks = (n_crops + chunk - 1) // chunk
for c in range(n_chunks):
    n = min(chunk, n_crops - c * chunk)
    hw = 48
    for bi in range(5):
        cs, hw = block_convs(bi, hw)
        if fused_l1 and bi < 2:
            cs = [(f"b{bi}.block 32->32->32 @{hw}", sum(fl for _, fl, _b in cs), 2 * 4 * 32 * hw * hw)]      # the intermediate stays in LDS
        if fused_s2 and bi == 2:
            cs = [(f"b2.conv1+ds 32->64 s2 @{hw}", cs[0][1] + cs[1][1], cs[0][2] + 4 * 64 * hw * hw), cs[2]]  # one pass over the input, two outputs
        seq += [(nm, fl * n, by * n) for nm, fl, by in cs]
hw = 24
for bi in range(5, 12):
    cs, hw = block_convs(bi, hw)
    seq += [(nm, fl * n_crops, by * n_crops) for nm, fl, by in cs]
seq.append(("proj 256->72", 2 * 256 * 72 * 36 * n_crops, 4 * (256 + 72) * 36 * n_crops))
s = n_crops // 2
head = [("fus0 144->108", 144 * 108, 144 + 108), ("fus1 108->72", 108 * 72, 108 + 72), ("fus2 72->72", 72 * 72, 72 + 72), ("tmp0 90->90", 8100, 180),
        ("tmp1 90->90", 8100, 180), ("tmp2 90->90", 8100, 180), ("reg0.conv1 76", 9 * 76 * 76, 2 * 76), ("reg0.conv2 76", 9 * 76 * 76, 3 * 76),
        ("reg1.conv1 76", 9 * 76 * 76, 2 * 76), ("reg1.conv2 76", 9 * 76 * 76, 3 * 76)]
seq += [(nm, 2 * mac * 36 * s, 4 * ch * 36 * s) for nm, mac, ch in head]

is_conv = lambda r: any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv3x3_c32_patch", "conv_split", "conv_block32", "conv_c64r", "conv_c64k", "conv_c32s2", "conv_w4"))


def label(name):
    args = name.split("<")[1].split(">")[0] if "<" in name else ""
    if "conv_split" in name:
        return "split f16 " + "x".join(args.split(", ")[:2])
    if "conv_w4" in name:
        a = args.split(", ")
        if len(a) > 4 and a[4] == "true":
            return f"split f16 phase planes -> {a[0]}x{a[1]} x 128 ch, 4 waves"
        return f"split f16 whole maps {a[0]}x{a[1]} x {'128' if a[3] == '1' else '64'} ch, 4 waves"
    if "conv_block32" in name:
        return "fused block 12x16 split f16"
    if "conv_c64r" in name:
        return "split f16 256x64 reg weights"
    if "conv_c32s2" in name:
        return "split f16 96px regW 3x3+1x1"
    if "conv_c64k" in name:
        return "split f16 128x64 regW K-split"
    if "conv3x3_c32_patch" in name:
        return "halo patch 16x24" + (" split f16" if args == "true" else "")
    return "fp32 " + "x".join(args.split(", ")[:2])


convs = [r for r in rows if is_conv(r)]
last = convs[-len(seq):]
agg = collections.OrderedDict()
for (nm, fl, by), r in zip(seq, last):
    tile = label(r["Kernel_Name"])
    a = agg.setdefault(nm, [0, 0.0, 0.0, tile, 0.0])
    a[0] += 1; a[1] += dur(r); a[2] += fl; a[4] += by
tot_t = sum(a[1] for a in agg.values()); tot_f = sum(a[2] for a in agg.values()); tot_b = sum(a[4] for a in agg.values())
print(f"{'conv':28s} {'kernel':28s} {'n':>3s} {'total us':>10s} {'TFLOP/s':>8s} {'GB':>6s} {'TB/s':>6s} {'% of conv time':>8s}")
for nm, (n, t, fl, tile, by) in agg.items():
    print(f"{nm:28s} {tile:28s} {n:3d} {t:10.1f} {fl/t/1e6:8.1f} {by/1e9:6.2f} {by/t/1e6:6.2f} {100*t/tot_t:8.2f}")
print(f"conv total {tot_t/1e3:.3f} ms, {tot_f/tot_t/1e6:.1f} TFLOP/s, {tot_b/1e9:.1f} GB of compulsory traffic = {tot_b/tot_t/1e6:.2f} TB/s")
t0 = int(last[0]["Start_Timestamp"])
others = collections.Counter()
for r in rows:
    if int(r["Start_Timestamp"]) >= t0 - 2e6 and not is_conv(r):
        others[r["Kernel_Name"].split("(")[0][-48:]] += dur(r)
print("other kernels in/around the last step (us):", {k: round(v, 1) for k, v in others.most_common(10)})
