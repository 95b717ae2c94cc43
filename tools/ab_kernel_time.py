"""Print the average duration of the kernels whose name contains the given substrings, from a rocprofv3 kernel_stats csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for pat in sys.argv[2:]:
    for r in rows:
        if pat in r["Name"]:
            out.append("%s=%.1fus" % (pat, float(r["AverageNs"]) / 1e3))
print(" ".join(out))
