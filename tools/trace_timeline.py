#!/usr/bin/env python3
"""Timeline of one step from a rocprofv3 --kernel-trace CSV: per-queue busy time, the union of the busy intervals, idle gaps, and the
launches in start order with the other queue's occupancy beside them.  What the per-kernel statistics cannot show: which launches
overlap, where the device idles, and which durations are inflated by waiting for CUs another stream's persistent kernels hold.
usage: trace_timeline.py <rocprof output dir> <marker kernel substring> [step index from the end, default 1] [--list] [--starved]"""
import csv, glob, os, re, sys


def load(d):
    f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)]
    if not f:
        raise SystemExit("no kernel_trace.csv under " + d)
    rows = list(csv.DictReader(open(f[0])))
    out = []
    for r in rows:
        q = r.get("Stream_Id") or r.get("Queue_Id") or "0"
        grid = (r.get("Grid_Size_X") or r.get("Grid_Size") or "", r.get("Grid_Size_Y") or "", r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "")
        out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), q, r["Kernel_Name"], grid))
    out.sort()
    return out


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:44]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    d, marker = sys.argv[1], sys.argv[2]
    back = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 1
    ev = load(d)
    marks = [i for i, e in enumerate(ev) if marker in e[3]]
    if len(marks) < back + 1:
        raise SystemExit("marker %r seen %d times" % (marker, len(marks)))
    a, b = marks[-back - 1], marks[-back]
    step = ev[a:b]
    t0, t1 = step[0][0], max(e[1] for e in step)
    print("step window: %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(step)))
    qs = sorted(set(e[2] for e in step))
    for q in qs:
        iv = [(e[0], e[1]) for e in step if e[2] == q]
        print("  queue %-6s %5d launches  busy %.3f ms (sum of durations %.3f)" % (q, len(iv), union(iv) / 1e6, sum(y - x for x, y in iv) / 1e6))
    allv = [(e[0], e[1]) for e in step]
    print("  device busy (union) %.3f ms, idle %.3f ms" % (union(allv) / 1e6, ((t1 - t0) - union(allv)) / 1e6))
    # idle gaps
    iv = sorted(allv)
    gaps, ce = [], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            gaps.append((s - ce, ce - t0))
        ce = max(ce, e)
    gaps.sort(reverse=True)
    print("  idle gaps: %d, > 2 us: %d (%.3f ms), > 10 us: %d (%.3f ms)" % (len(gaps), sum(g > 2000 for g, _ in gaps), sum(g for g, _ in gaps if g > 2000) / 1e6,
                                                                         sum(g > 10000 for g, _ in gaps), sum(g for g, _ in gaps if g > 10000) / 1e6))
    for g, at in gaps[:8]:
        print("     %.1f us at +%.3f ms" % (g / 1e3, at / 1e6))
    # time where both queues run
    if len(qs) > 1:
        per = [union([(e[0], e[1]) for e in step if e[2] == q]) for q in qs]
        print("  overlap (sum of per-queue busy - union): %.3f ms" % ((sum(per) - union(allv)) / 1e6))
    if "--starved" in sys.argv:
        # launches that took far longer than the same kernel's fastest launch of the same grid in this trace while another queue was busy:
        # a whole-CU persistent kernel on another stream admits nothing beside it, so a small kernel launched behind it "runs" until it ends.
        # CANDIDATES only: a persistent kernel launches the same grid for every layer size, so its rows compare different problems - read
        # them against the layer; for the small per-layer kernels (finalisations, scale vectors, sums) the comparison is like for like.
        best = {}
        for s, e, q, n, g in ((x[0], x[1], x[2], x[3], x[4] if len(x) > 4 else None) for x in ev):
            k = (n, g)
            best[k] = min(best.get(k, 1 << 62), e - s)
        rows = []
        for x in step:
            s, e, q, n = x[:4]
            g = x[4] if len(x) > 4 else None
            d, b = e - s, best[(n, g)]
            if d > 3 * b and d - b > 50000:
                other = sorted(set(short(y[3])[:28] for y in step if y[2] != q and y[0] < e and y[1] > s))
                rows.append((d - b, s, d, b, q, n, other))
        rows.sort(reverse=True)
        print("  starved launches (> 3x the kernel's fastest launch of that grid, > 50 us lost): %d, %.3f ms lost" % (len(rows), sum(r[0] for r in rows) / 1e6))
        for lost, s, d, b, q, n, other in rows[:25]:
            print("%9.3f %8.1f us (fastest %7.1f)  q%-4s %-40s | %s" % ((s - t0) / 1e6, d / 1e3, b / 1e3, q, short(n)[:40], ", ".join(other)[:70]))
    if "--list" in sys.argv:
        for s, e, q, n, _g in step:
            other = [x for x in step if x[2] != q and x[0] < e and x[1] > s]
            print("%9.3f %8.1f us  q%-4s %-44s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, short(n), ("| " + ", ".join(sorted(set(short(x[3])[:24] for x in other)))[:70]) if other else ""))


if __name__ == "__main__":
    main()
