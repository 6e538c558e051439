#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes by kernel name and grid size: mean duration and mean of every counter per dispatch.
Usage: tools/pmc_byname.py <dir with pmc*/ subdirectories> [min_grid]"""
import csv, glob, sys, collections
root = sys.argv[1]; min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in sorted(glob.glob(root + '/pmc*/**/*_counter_collection.csv', recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        if 'nbls' not in r['Kernel_Name'] or int(r['Grid_Size']) < min_grid: continue
        key = (r['Kernel_Name'], int(r['Grid_Size']))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        d = (f, r['Dispatch_Id'])
        if d not in seen: seen.add(d); dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
names = sorted({c for v in acc.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'grid', 'dispatches', 'avg_us_under_pmc'] + names + ['VALU_per_wave'])
for key in sorted(acc):
    v = acc[key]
    row = [key[0], key[1], len(dur[key]), round(sum(dur[key]) / len(dur[key]) / 1e3, 1)] + [round(sum(v[c]) / len(v[c]), 1) if c in v else '' for c in names]
    row.append(round(sum(v['SQ_INSTS_VALU']) / sum(v['SQ_WAVES']), 1) if 'SQ_INSTS_VALU' in v and 'SQ_WAVES' in v and sum(v['SQ_WAVES']) else '')
    w.writerow(row)
