#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command, csv output) to HBM bytes per
launch of every kernel, in the form bench.py's `roofline.traffic` reads (profiles/<round>_pmc_traffic.json).

usage: pmc_traffic_all.py <fetch_dir> <write_dir> <round tag, e.g. r03> <out.json> [command that was profiled]

FETCH_SIZE / WRITE_SIZE are in KB (x1024).  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide coalesced streaming read -> `fetch_bytes_x2`; streams of 64-B runs are counted exactly (round 6 calibration on known
byte counts: tools/calib/, profiles/<round>_fetch_calib.txt), so both figures are kept and bench.py picks per kernel family."""
import collections, csv, glob, json, re, sys

fd, wd, rnd, out = sys.argv[1:5]
cmd = sys.argv[5] if len(sys.argv) > 5 else ""


def load(d, counter):
    per = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()
            key = (f, r["Dispatch_Id"])
            per[k][key] = per[k].get(key, 0.0) + float(r["Counter_Value"])
    return per


fa, wa = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
rows = []
for k, d in fa.items():
    n = len(d)
    fr = sum(d.values()) * 1024 / n
    wr = (sum(wa[k].values()) * 1024 / max(len(wa[k]), 1)) if k in wa else 0.0
    rows.append(dict(kernel=k, launches=n, fetch_bytes_raw=fr, fetch_bytes_x2=2 * fr, write_bytes=wr,
                     fetch_bytes_raw_max_launch=max(d.values()) * 1024))
rows.sort(key=lambda r: -(r["fetch_bytes_x2"] + r["write_bytes"]) * r["launches"])
# calibration: tools/calib/run_fetch_calib.sh (known byte counts per access pattern) -> profiles/<round>_fetch_calib.json; bench.py applies it.
# (Rounds 2-5 calibrated on the 64^3 x 32-channel pool launch "reading 2.147 GB once" - wrong in inference, where that launch reads a
#  brick-sparse tensor and skips ~85 % of it: the 0.138 figure of profiles/r05_pmc_traffic.json says nothing about the counter.)
cal = dict(note="FETCH_SIZE = 64 B per fabric read request: x2 for contiguous streams (128-B requests), x1 for 64-B runs; see <round>_fetch_calib.json")
json.dump(dict(round=rnd, command=cmd, method=__doc__, calibration=cal, kernels=rows[:60]), open(out, "w"), indent=1)
for r in rows[:16]:
    print("%-46s n=%4d fetch raw %8.1f MB (x2 %8.1f)  write %8.1f MB" % (r["kernel"][:46], r["launches"], r["fetch_bytes_raw"] / 1e6,
                                                                        r["fetch_bytes_x2"] / 1e6, r["write_bytes"] / 1e6))

