#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel totals and, for the conv_igemm
kernel, per-(grid, instantiation) average durations of the last step.  Usage:
    python tools_layer_profile.py <kernel_trace.csv> [launches_per_step]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-60:]
    tot[k][0] += 1
    tot[k][1] += dur(r)
print("kernel totals (whole run):")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t/1e3:10.3f} ms  {n:6d} calls  {t/n:9.1f} us avg  {k}")
convs = [r for r in rows if "conv_igemm" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if n:
    last = convs[-n:]
    groups = collections.OrderedDict()
    for r in last:
        key = (r["Kernel_Name"].split("<")[1].split(">")[0], r["Grid_Size_X"], r["Grid_Size_Y"])
        groups.setdefault(key, []).append(dur(r))
    print("last step, conv_igemm by (tile, grid):")
    for k, v in groups.items():
        print(f"  {k}: {len(v):3d} launches, avg {sum(v)/len(v):8.1f} us, total {sum(v)/1e3:7.3f} ms")
    print(f"  step conv total {sum(dur(r) for r in last)/1e3:.3f} ms")
