#!/bin/bash
# Kernel timeline of ONE nbls_msm_dev call (65,536 G1 points, 255-bit scalars) under rocprofv3 --kernel-trace.  Usage: tools/msm_timeline.sh [n]
export TMPDIR=/tmp
n=${1:-65536}
out=$PWD/gpurun_out/msm_timeline; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/exp_msm.py $n 0 255 3 > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $out/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tail = rows[-400:]
gaps = [(int(tail[i]['Start_Timestamp']) - int(tail[i - 1]['End_Timestamp']), i) for i in range(1, len(tail))]
cut = max(gaps)[1] if gaps else 0
last = tail[cut:]
t0 = int(last[0]['Start_Timestamp'])
print('kernels of the last call: %d, span %.3f ms, sum of durations %.3f ms' % (len(last), (max(int(r['End_Timestamp']) for r in last) - t0) / 1e6, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6))
for r in last:
    print('%9.3f ms  +%7.3f ms  grid %-9s %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('Grid_Size', r.get('Grid_Size_X', '?')), r['Kernel_Name'][:60]))
PY
