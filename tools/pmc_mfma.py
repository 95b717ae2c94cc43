#!/usr/bin/env python3
"""Reduce one rocprofv3 --pmc pass (SQ / GRBM counters; rocpd sqlite output) to per-kernel matrix-core utilisation.

usage: pmc_mfma.py <results.db> <out.json>
Counters of the pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_WAVE_CYCLES
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE.
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of the SIMD-cycles of CUs that held a wave on which
               the matrix pipe was executing (rocprofv3 sums both over the chip)
  clock_ghz  = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md 'DVFS give-back': reads high below ~0.3 ms)
  frac_of_f16_peak = mfma_busy x clock / 2.4 GHz  (issued MFMA rate against the 2.5 PFLOP/s dense peak)"""
import json, re, sqlite3, sys
from collections import defaultdict

dbp, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(dbp).cursor()
acc = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set); dur = defaultdict(float)
for name, did, cname, val, d in cur.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
    k = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", "")).strip()
    acc[k][cname] += val
    if did not in disp[k]:
        dur[k] += d
        disp[k].add(did)
rows = []
for k, c in acc.items():
    n = len(disp[k])
    r = dict(kernel=k, launches=n, avg_us=dur[k] / n / 1e3, **{a: b / n for a, b in c.items()})
    if c.get("SQ_BUSY_CU_CYCLES"):
        r["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_BUSY_CU_CYCLES"])
    if c.get("GRBM_GUI_ACTIVE") and dur[k]:
        r["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / dur[k]
        if "mfma_busy" in r:
            r["frac_of_f16_peak"] = r["mfma_busy"] * r["clock_ghz"] / 2.4
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_F16"):
        r["valu_insts_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_VALU_MFMA_MOPS_F16"] / 512 * 0) / max(c["SQ_INSTS_VALU_MFMA_MOPS_F16"] / 512.0, 1.0)
    rows.append(r)
rows.sort(key=lambda r: -r["avg_us"] * r["launches"])
json.dump(dict(method=__doc__, kernels=rows[:40]), open(out, "w"), indent=1)
for r in rows[:14]:
    print("%-46s n=%4d avg=%8.1f us  mfma_busy=%.3f  clock=%.2f GHz  of_f16_peak=%.3f" % (
        r["kernel"][:46], r["launches"], r["avg_us"], r.get("mfma_busy", 0), r.get("clock_ghz", 0), r.get("frac_of_f16_peak", 0)))
