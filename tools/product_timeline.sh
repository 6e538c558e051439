#!/bin/bash
# Kernel timeline of ONE Miller product (tools/product_time.py n 1) under rocprofv3 --kernel-trace: start offset, duration and hardware queue of every kernel of the last call.
# Usage: tools/product_timeline.sh [n]
export TMPDIR=/tmp
n=${1:-262144}
out=$PWD/gpurun_out/product_timeline_n$n; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/product_time.py $n 1 > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $out/timeline.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Kernel_Name'].startswith('nbls') or 'copyBuffer' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
gaps = [(int(rows[i]['Start_Timestamp']) - max(int(x['End_Timestamp']) for x in rows[:i]), i) for i in range(1, len(rows))]
cut = max(gaps)[1] if gaps else 0     # the last call starts after the largest idle gap
last = rows[cut:]
t0 = int(last[0]['Start_Timestamp'])
print('kernels of the last call: %d, span %.3f ms, sum of durations %.3f ms' % (len(last), (max(int(r['End_Timestamp']) for r in last) - t0) / 1e6, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6))
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d >= 0.02: print('%9.3f ms  +%7.3f ms  queue %-3s grid %-8s %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, d, r.get('Queue_Id', '?'), r.get('Grid_Size', r.get('Grid_Size_X', '?')), r['Kernel_Name'][:40]))
PY
