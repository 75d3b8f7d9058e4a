#!/usr/bin/env python3
"""HBM traffic of the conv_igemm kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share
a pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Corrections per that guide's HBM section:
counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read, so it is
doubled; WRITE_SIZE is exact for 16 B/lane stores.
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [label] [n_steps] [fp32|split_f16]
The last argument names the arithmetic the profiled command ran (ut_set_conv_arithmetic): the per-launch figure is over the
kernels bench.py's roofline object covers in that mode.
With n_steps (hot-path steps the profiled command ran, warm-up included) the summary also holds the HBM bytes of one
whole step over ALL of the library's kernels (resampler, stem, convolutions, head glue, FK) and their ratio to the
algorithmic bytes of SURVEY.md 8(d) (2048 hand-frames x 74,220 B)."""
import csv
import json
import sys


KIND = sys.argv[6] if len(sys.argv) > 6 else "fp32"


def covered(name):
    if KIND == "split_f16":
        return ("conv_split_kernel" in name or "conv3x3_c32_patch_kernel<true>" in name or "conv_block32_kernel" in name
                or "conv_c64r_kernel" in name or "conv_c64k_kernel" in name or "conv_w4_kernel" in name or "conv_c32s2_kernel" in name)
    return "conv_igemm" in name or "conv3x3_c32_patch" in name


def per_launch(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if covered(r["Kernel_Name"]) and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


def per_kernel(path, counter):
    tot = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] == counter and ("ut::" in k):
            name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"])
    return tot


fetch, nf = per_launch(sys.argv[1], "FETCH_SIZE")
write, nw = per_launch(sys.argv[2], "WRITE_SIZE")
out = {
    "conv_arithmetic": KIND,
    "kernel": ("conv_w4_kernel (stride-1 3x3 of layer2 .. layer4 and of the pose regressor, stride-2 entries of layer3 / layer4 as phase planes) + conv_c32s2_kernel (layer2 entry: 3x3 / 2 + shortcut) + conv_block32_kernel (layer1, one launch per BasicBlock)" if KIND == "split_f16" else
               "conv_igemm_kernel (all instantiations) + conv3x3_c32_patch_kernel"), "label": sys.argv[4] if len(sys.argv) > 4 else "",
    "launches_fetch_pass": nf, "launches_write_pass": nw,
    "FETCH_SIZE_KiB_per_launch_raw": fetch / max(nf, 1), "WRITE_SIZE_KiB_per_launch": write / max(nw, 1),
    "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
    "traffic_bytes_per_launch": (2.0 * fetch / max(nf, 1) + write / max(nw, 1)) * 1024.0,
}
if len(sys.argv) > 5:
    n_steps = int(sys.argv[5])
    fk, wk = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    per = {k: (2.0 * fk.get(k, 0.0) + wk.get(k, 0.0)) * 1024.0 / n_steps for k in sorted(set(fk) | set(wk))}
    step = sum(per.values())
    algo = 2048 * 74220.0
    out.update({"steps_in_run": n_steps, "step_traffic_bytes_all_kernels": step,
                "step_traffic_bytes_by_kernel": {k: round(v) for k, v in sorted(per.items(), key=lambda kv: -kv[1])},
                "algorithmic_bytes_per_step": algo, "ratio_to_algorithmic": step / algo})
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
