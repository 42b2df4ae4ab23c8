#!/usr/bin/env python
"""HBM-side traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

    python tools/gpu/gpu_pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring> <out.json>

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of 16 B/lane streaming reads at 64 bytes, so it is doubled.
WRITE_SIZE is uncalibrated there and taken as is.  Infinity-Cache hits are counted (fabric-side counters)."""
import csv, json, sys


def avg(path, counter, needle):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch_csv, write_csv, needle, out = sys.argv[1:5]
f, nf = avg(fetch_csv, "FETCH_SIZE", needle)
w, nw = avg(write_csv, "WRITE_SIZE", needle)
res = {"kernel_substring": needle, "launches_fetch_pass": nf, "launches_write_pass": nw,
       "FETCH_SIZE_KiB_raw_per_launch": f, "WRITE_SIZE_KiB_raw_per_launch": w,
       "fetch_bytes_corrected_per_launch": None if f is None else 2.0 * f * 1024.0,
       "write_bytes_per_launch": None if w is None else w * 1024.0}
if f is not None and w is not None:
    res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected_per_launch"] + res["write_bytes_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
