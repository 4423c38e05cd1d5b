#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output into the small files committed under profiles/.

  summarise_profile.py stats <kernel_stats.csv> <out.csv>          short kernel names, top rows
  summarise_profile.py pmc <out.json> <counter_collection.csv>...  mean counter per kernel; derives
                                                                  HBM bytes per launch with the
                                                                  gfx950 FETCH_SIZE x2 correction
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '')
    return name[-60:]


def stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for r in rows[:25]:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'],
                        r['MinNs'], r['MaxNs']])


def pmc(dst, files):
    d = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k.startswith('at::') or 'rocclr' in k or 'rocblas' in k:
                continue
            d[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    out = collections.defaultdict(dict)
    for (k, c), v in d.items():
        out[k][c] = sum(v) / len(v)
        out[k]['launches_' + c] = len(v)
    res = {'per_kernel_mean': out}
    for k, c in out.items():
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
            # coalesced streaming reads (MI355X_MICROARCH.md, HBM section): double it.  Calibration in
            # this code base: glm_pass_kernel FETCH_SIZE x 2 x 1024 = 8.2 GB = its algorithmic bytes.
            c['hbm_bytes_per_launch'] = (2.0 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0
    w = [k for k in out if k.startswith('wsyrk_glds_kernel') or k.startswith('wsyrk_kernel')]
    if w and 'hbm_bytes_per_launch' in out[w[0]]:
        res['wsyrk_hbm_bytes_per_launch'] = out[w[0]]['hbm_bytes_per_launch']
    json.dump(res, open(dst, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3:])
