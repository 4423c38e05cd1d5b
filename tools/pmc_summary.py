"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel (short names)."""
import csv, collections, glob, sys
pat = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else ''
d = collections.defaultdict(list)
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        if only and only not in k:
            continue
        d[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()):
    print('%-42s %-28s n=%3d mean=%.6g' % (k, c, len(v), sum(v) / len(v)))
