import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "hvp_multi_kernel" in r["Kernel_Name"]]
live = [x for x in d if x > 100]
print(len(d), len(live), "mean %.1f us  median %.1f  min %.1f  max %.1f" % (statistics.mean(live), statistics.median(live), min(live), max(live)))
