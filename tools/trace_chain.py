import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "chain" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = [(("X" if "<true>" in r["Kernel_Name"] else ("c" if "prior" in r["Kernel_Name"] else "p")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
print(" ".join("%s%.0f" % kd for kd in out[:9]), "...", " ".join("%s%.0f" % kd for kd in out[-9:]))
