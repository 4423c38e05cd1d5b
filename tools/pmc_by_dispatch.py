"""Prints a counter per dispatch (in order) for kernels whose name contains a substring."""
import csv, glob, sys
pat, sub, counter = sys.argv[1], sys.argv[2], sys.argv[3]
rows = []
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name'] and r['Counter_Name'] == counter:
            rows.append((int(r['Dispatch_Id']), float(r['Counter_Value']), int(r['Grid_Size'])))
rows.sort()
for d, v, g in rows:
    print(d, g // 256, v)
