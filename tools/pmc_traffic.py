#!/usr/bin/env python3
"""HBM traffic of the conv_igemm kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share
a pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Corrections per that guide's HBM section:
counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read, so it is
doubled; WRITE_SIZE is exact for 16 B/lane stores.
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [label]"""
import csv
import json
import sys


def per_launch(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if ("conv_igemm" in r["Kernel_Name"] or "conv3x3_c32_patch" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fetch, nf = per_launch(sys.argv[1], "FETCH_SIZE")
write, nw = per_launch(sys.argv[2], "WRITE_SIZE")
out = {
    "kernel": "conv_igemm_kernel (all instantiations) + conv3x3_c32_patch_kernel", "label": sys.argv[4] if len(sys.argv) > 4 else "",
    "launches_fetch_pass": nf, "launches_write_pass": nw,
    "FETCH_SIZE_KiB_per_launch_raw": fetch / max(nf, 1), "WRITE_SIZE_KiB_per_launch": write / max(nw, 1),
    "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
    "traffic_bytes_per_launch": (2.0 * fetch / max(nf, 1) + write / max(nw, 1)) * 1024.0,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
