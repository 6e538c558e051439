#!/bin/bash
# Kernel timeline of ONE verify (n = 1) / one verifyBatch: rocprofv3 --kernel-trace of tools/verify_breakdown.py, the kernels of the last call listed with
# start offset, duration and hardware queue -- shows whether the key-decode, signature-decode and hash chains of nbls_verify_batch really overlap.
# Usage: tools/verify_timeline.sh [n] [tag]
export TMPDIR=/tmp
n=${1:-1}; tag=${2:-n$n}
out=$PWD/gpurun_out/verify_timeline_$tag; rm -rf $out; mkdir -p $out
NBLS_VB_NOTIMING=1 rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/verify_breakdown.py $n > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $out/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last call: kernels after the last gap of more than 0.5 ms... verify_breakdown runs the timed call last; take the trailing 60 kernels and cut at the largest gap
tail = rows[-120:]
gaps = [(int(tail[i]['Start_Timestamp']) - int(tail[i - 1]['End_Timestamp']), i) for i in range(1, len(tail))]
cut = max(gaps)[1] if gaps else 0
last = tail[cut:]
t0 = int(last[0]['Start_Timestamp'])
print('kernels of the last call: %d, span %.3f ms, sum of durations %.3f ms' % (len(last), (max(int(r['End_Timestamp']) for r in last) - t0) / 1e6, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6))
for r in last:
    print('%9.3f ms  +%7.3f ms  queue %-3s grid %-8s %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('Queue_Id', '?'), r.get('Grid_Size', r.get('Grid_Size_X', '?')), r['Kernel_Name'][:40]))
PY
