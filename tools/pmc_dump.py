#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database.  usage: pmc_dump.py <results.db> [kernel substring]"""
import re, sqlite3, sys
from collections import defaultdict
cur = sqlite3.connect(sys.argv[1]).cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set); dur = defaultdict(float)
for name, did, cname, val, d in cur.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
    k = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", "")).strip()
    if sub not in k: continue
    acc[k][cname] += val
    if did not in disp[k]: dur[k] += d; disp[k].add(did)
for k, c in acc.items():
    n = len(disp[k])
    print("%s  n=%d avg=%.1f us" % (k, n, dur[k] / n / 1e3))
    for a in sorted(c): print("    %-32s %.4g" % (a, c[a] / n))
