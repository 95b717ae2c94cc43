"""Per (kernel, grid) launch durations of a rocprofv3 --kernel-trace CSV: count, median, min.  usage: summarize_trace.py <dir> <regex>"""
import csv, glob, collections, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = re.compile(sys.argv[2])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    if pat.search(n):
        agg[(n, r.get("Grid_Size_X") or r.get("Grid_Size") or "", r.get("Grid_Size_Y") or "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v = sorted(v)
    print("%-52s grid %-9s %-4s n=%3d  median %8.1f us  min %8.1f" % (k[0][:52], k[1], k[2], len(v), v[len(v) // 2], v[0]))
