#!/usr/bin/env python3
"""Per-call summary of rocprofv3 PMC passes over `tools/exp_time.py <batch> <reps>` (2 warm-up + reps timed + 1 timing call = reps + 3 identical pairing calls):
per kernel name -- launches per call, wavefronts, VALU instructions per wavefront, mean duration, share of LDS cycles lost to bank conflicts, issue slots used --
and per call: VALU wave-instructions, HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, in KB as the counters report them; FETCH_SIZE doubled per the gfx950 note of
MI355X_MICROARCH.md).  Usage: tools/pmc_percall.py <dir with pmc*/> <batch> <calls> [min_grid]  -> JSON on stdout"""
import csv, glob, json, sys, collections
root, batch, calls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); min_grid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list); ndisp = collections.defaultdict(set)
for f in sorted(glob.glob(root + '/pmc*/**/*_counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'nbls' not in r['Kernel_Name'] or int(r['Grid_Size']) < min_grid: continue
        k = r['Kernel_Name']
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        d = (f, r['Dispatch_Id'])
        if d not in ndisp[k]: ndisp[k].add(d); dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
npass = len(glob.glob(root + '/pmc*/'))
mean = lambda v: sum(v) / len(v) if v else 0.0
progs = {}; tot = collections.Counter()
for k in sorted(acc):
    v = acc[k]
    per_call = len(dur[k]) / npass / calls
    waves = mean(v['SQ_WAVES']); valu = mean(v['SQ_INSTS_VALU']); us = mean(dur[k]) / 1e3
    progs[k] = {'launches_per_call': round(per_call, 2), 'waves': round(waves), 'valu_per_wave': round(valu / waves) if waves else 0, 'avg_us_under_pmc': round(us, 1),
                'issue_slots_used': round(valu * 4 / (us * 1e-6 * 2.4e9 * 1024), 3) if us else 0,
                'lds_conflict_frac': round(mean(v['SQ_LDS_BANK_CONFLICT']) / mean(v['SQ_LDS_IDX_ACTIVE']), 3) if mean(v['SQ_LDS_IDX_ACTIVE']) else None,
                'wait_any_frac': round(mean(v['SQ_WAIT_ANY']) / mean(v['SQ_WAVE_CYCLES']), 3) if mean(v['SQ_WAVE_CYCLES']) and v['SQ_WAIT_ANY'] else None}
    tot['valu'] += valu * per_call; tot['us'] += us * per_call
    tot['fetch'] += mean(v['FETCH_SIZE']) * 1024 * per_call; tot['write'] += mean(v['WRITE_SIZE']) * 1024 * per_call
ALG = batch * (288 + 832 + 128 + 832 + 768 + 5 * 1536 + 2 * 2304 + 5376 + 576 + 2 * 26112)   # DESIGN.md section 4, two-program Miller loop
out = {'batch': batch, 'calls_profiled': calls, 'bytes_per_step': round(2 * tot['fetch'] + tot['write']), 'fetch_size_bytes_raw': round(tot['fetch']), 'write_size_bytes': round(tot['write']),
       'algorithmic_bytes_per_step': ALG, 'valu_wave_instructions_per_step': round(tot['valu']), 'valu_wave_instructions_per_pairing': round(tot['valu'] / batch * 64 / 64, 1),
       'valu_issue_busy': round(tot['valu'] * 4 / (tot['us'] * 1e-6 * 2.4e9 * 1024), 4), 'kernel_us_per_call_under_pmc': round(tot['us'], 1), 'kernels': progs}
json.dump(out, sys.stdout, indent=1)
