#!/bin/bash
# FETCH_SIZE per access pattern on known byte counts (through gpurun): bash tools/calib/run_fetch_calib.sh > gpurun_out/fetch_calib.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
hipcc -O3 -w --offload-arch=gfx950 tools/calib/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib 2
rm -rf /tmp/fc; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -- /tmp/fetch_calib 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, json
per = collections.defaultdict(list)
for f in glob.glob("/tmp/fc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "read_kernel<" in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024)
known = 2 * (1 << 30)
names = {"0": "contig16", "1": "pair32", "2": "seg64_256", "3": "seg64_128", "4": "dword"}
out = {}
for k, v in sorted(per.items()):
    p = k.split("<")[1].split(">")[0]
    raw = sum(v) / len(v)
    out[names[p]] = dict(launches=len(v), fetch_size_bytes_raw=raw, known_bytes=known, raw_over_known=raw / known, factor=known / raw)
    print("%-10s FETCH_SIZE raw %.3f GB over %d launches; known %.3f GB; raw/known %.3f -> multiply the counter by %.2f" % (names[p], raw / 1e9, len(v), known / 1e9, raw / known, known / raw))
json.dump(out, open("gpurun_out/fetch_calib.json", "w"), indent=1)
PY
