"""Kernel timeline of the last few steps of a run: name, duration, gap to the previous kernel (from a rocprofv3 kernel trace)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sel = rows[-(n + skip):len(rows) - skip] if skip else rows[-n:]
prev = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s dur %8.1f us gap %7.1f us" % (r["Kernel_Name"][:44], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0)); prev = e
