export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_r6b; rm -rf $out; mkdir -p $out
common="--no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/b4096_inflight12 -- python bench.py --steps 192 --warmup 12 --batch 4096 --inflight 12 --mark-timed-region $common > $out/b4096_inflight12.log 2>&1
f=$(find $out/b4096_inflight12 -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_b4096_inflight12.csv
python - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
f = glob.glob(out + '/b4096_inflight12/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = sorted({r['Kernel_Name'][:60] for r in rows if 'fill' in r['Kernel_Name'].lower()})
print(names)
marks = [i for i, r in enumerate(rows) if 'fillfunctor' in r['Kernel_Name'].lower()]
assert len(marks) >= 2, 'markers not found'
timed = [r for r in rows[marks[0] + 1:marks[1]] if r['Kernel_Name'].startswith('nbls_')]
span = (max(int(r['End_Timestamp']) for r in timed) - min(int(r['Start_Timestamp']) for r in timed)) / 1e9
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in timed) / 1e9
steps = 192
json.dump({'what': 'nbls kernels between the two marker kernels of `python bench.py --steps 192 --warmup 12 --batch 4096 --inflight 12 --mark-timed-region` under rocprofv3 --kernel-trace: the timed region of `value`',
           'timed_calls': steps, 'kernels': len(timed), 'kernels_per_call': round(len(timed) / steps, 2), 'span_s': round(span, 6), 'sum_of_kernel_durations_s': round(busy, 6),
           'mean_kernels_in_flight': round(busy / span, 2), 'pairings_per_s_over_span': round(steps * 4096 / span, 1),
           'frac_at_value_from_span': round(steps * 4096 / span * 19722 * 300 / 1e12 / (256 * 64 * 2.4e9 / 1e12), 4)},
          open(out + '/inflight12_span.json', 'w'), indent=1)
print(open(out + '/inflight12_span.json').read())
PY
rm -rf $out/b4096_inflight12
