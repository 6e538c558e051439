#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile_round.sh for one bench.py run (pairing batch only).

Every bench step launches the same sequence of kernels (nbls_vm_kernel running the step programs miller_fe, fe_easy, expx,
fe_mid1, expx, expx, expx, fe_mid2, expx, fe_final, and nbls_fp_inv_kernel after the Miller programs lines_pq + acc_fe), so dispatches are attributed
to step programs by their position in that cycle.  Prints CSV: per program, dispatch count, mean duration and the mean
of every collected counter per dispatch.  Usage: tools/pmc_summary.py <gpurun_out/prof_TAG> [min_grid]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
CYCLE = ['lines_pq', 'acc_fe', 'fe_easy', 'expx', 'fe_mid1', 'expx', 'expx', 'expx', 'fe_mid2', 'expx', 'fe_final']
if os.environ.get('NBLS_FUSED_MILLER'):
    CYCLE = ['miller_fe'] + CYCLE[2:]   # the one-program Miller loop of round 1 (A/B switch of the library)
VM = ('nbls_vm_kernel', 'nbls_vm_kernel_fair', 'nbls_vm_kernel_sc', 'nbls_vm_kernel_fair_sc', 'nbls_vm_kernel_ls4')   # the launcher picks an instantiation by launch shape (csrc/vm_kernel.hip)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(root + '/pmc*/**/*_counter_collection.csv', recursive=True)):
    # min_grid drops the small launches of bench.py's parity spot check (grid = 64 lanes x waves; the inversion kernel runs one lane per item)
    rows = [r for r in csv.DictReader(open(f)) if 'nbls' in r['Kernel_Name'] and int(r['Grid_Size']) >= (min_grid if r['Kernel_Name'] in VM else min_grid // 16)]
    ids = sorted({int(r['Dispatch_Id']) for r in rows if r['Kernel_Name'] in VM})
    label = {d: CYCLE[i % len(CYCLE)] for i, d in enumerate(ids)}
    seen = set()
    for r in rows:
        d = int(r['Dispatch_Id'])
        key = label[d] if r['Kernel_Name'] in VM else 'fp_inv'
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if d not in seen:
            seen.add(d); dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
names = sorted({c for v in acc.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(['program', 'dispatches', 'avg_us_under_pmc'] + names)
for key in ['miller_fe', 'lines_pq', 'acc_fe', 'fp_inv', 'fe_easy', 'expx', 'fe_mid1', 'fe_mid2', 'fe_final']:
    if key not in acc: continue
    v = acc[key]
    w.writerow([key, len(dur[key]), round(sum(dur[key]) / len(dur[key]) / 1e3, 1)] + [round(sum(v[c]) / len(v[c]), 1) if c in v else '' for c in names])
