#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per (kernel, grid, LDS size) group, mean counter value per dispatch.
Usage: tools/pmc_summary.py <dir with pmc*/.../*_counter_collection.csv> [> summary.csv]"""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(root + '/pmc*/**/*_counter_collection.csv', recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        if 'nbls' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'], int(r['Grid_Size']), int(r['LDS_Block_Size']))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        d = (f, r['Dispatch_Id'])
        if d not in seen:
            seen.add(d); dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
names = sorted({c for v in acc.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'grid', 'lds', 'dispatches', 'avg_us_under_pmc'] + names)
for key in sorted(acc, key=lambda k: -sum(dur[k])):
    v = acc[key]
    w.writerow([key[0], key[1], key[2], len(dur[key]), round(sum(dur[key]) / len(dur[key]) / 1e3, 1)] + [round(sum(v[c]) / len(v[c]), 1) if c in v else '' for c in names])
