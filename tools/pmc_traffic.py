#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) to HBM bytes per launch of one kernel.

usage: pmc_traffic.py <fetch_dir> <write_dir> <kernel substring> <label> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of a wide coalesced read stream, so FETCH_SIZE is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, glob, json, sys

def per_launch(d, counter, sub):
    vals = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or sub not in r["Kernel_Name"]:
                continue
            key = (r.get("Dispatch_Id"), f)
            vals[key] = vals.get(key, 0.0) + float(r["Counter_Value"])
    return list(vals.values())

fd, wd, sub, label, out = sys.argv[1:6]
f = per_launch(fd, "FETCH_SIZE", sub); w = per_launch(wd, "WRITE_SIZE", sub)
assert f and w, "no launches of %r in the counter files" % sub
fk = sum(f) / len(f); wk = sum(w) / len(w)
rec = dict(kernel=label, launches=len(f), fetch_size_kb_raw_per_launch=fk, write_size_kb_per_launch=wk,
           fetch_bytes_per_launch_corrected=fk * 1024 * 2, write_bytes_per_launch=wk * 1024,
           traffic_bytes_per_launch=fk * 1024 * 2 + wk * 1024,
           method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 "
                  "--no-cpu-baseline`; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md "
                  "HBM section); KB -> bytes x1024; average over every launch of the kernel in the profiled steps")
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
