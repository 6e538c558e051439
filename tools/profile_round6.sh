#!/bin/bash
# Round-6 profile set (GPU box, through gpurun; about four minutes).  Everything lands in gpurun_out/prof_r6/; tools/collect_round6.sh copies what is kept to
# profiles/round6_*.  Counters are collected in their own passes with --kernel-trace only.
#   1. rocprofv3 --kernel-trace --stats: one 4096-pairing call at a time / twelve in flight (the configuration of `value`, with the span of the timed region) /
#      one 65,536-pairing call at a time                                        -> kernel_stats_b4096.csv, kernel_stats_b4096_inflight12.csv + inflight12_span.json, kernel_stats_b65536.csv
#   2. PMC passes over tools/exp_time.py at 4096 and 65,536 (kernels by name; NBLS_HALVES_MIN=0: kernels running alone)   -> pmc_b4096.csv / .json, pmc_b65536.csv / .json, hbm_traffic.json
#   3. verifyBatch: per-kernel times (timing mode) and the kernel timeline of one call as it runs in production          -> verify_breakdown_n65536.txt, verify_breakdown_n1.txt, verify_timeline_n65536.txt, verify_timeline_n1.txt
#   4. A/B texts of the round's changes on this box (one-limb-per-lane powers on / off, the one-limb-per-lane interpreter, Miller form against size, sign sizes), clocks under load, stress parity
#   5. the JSON line of a default bench run and of the driver's arguments      -> bench_default.json, bench_driver_args.json
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_r6; rm -rf $out; mkdir -p $out
common="--no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
stats() {   # <name> <bench args>
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1 -- python bench.py $2 $common > $out/$1.log 2>&1
  f=$(find $out/$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_$1.csv
}
export NBLS_HALVES_MIN=0
stats b4096 "--steps 16 --warmup 2 --batch 4096 --inflight 1"
stats b65536 "--steps 3 --warmup 1 --batch 65536 --inflight 1"
unset NBLS_HALVES_MIN
stats b4096_inflight12 "--steps 192 --warmup 12 --batch 4096 --inflight 12 --mark-timed-region"
python - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
f = glob.glob(out + '/b4096_inflight12/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'fillfunctor' in r['Kernel_Name'].lower()]      # torch's fill kernels (the runtime's own fillBuffer kernels -- hipMemsetAsync, e.g. the key wipes of round 6 -- also carry 'fill' in their names)
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
PY
export NBLS_HALVES_MIN=0 NBLS_FUSED_MILLER=0
for b in 4096 65536; do
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_b$b/pmc$i -- python tools/exp_time.py $b 2 > $out/pmc_b${b}_$i.log 2>&1
  done
  python tools/pmc_byname.py $out/pmc_b$b $((b / 16)) > $out/pmc_b$b.csv
  python tools/pmc_percall.py $out/pmc_b$b $b 5 $((b / 16)) > $out/pmc_b$b.json
done
unset NBLS_HALVES_MIN NBLS_FUSED_MILLER
python - $out <<'PY'
import json, sys
out = sys.argv[1]
a, b = json.load(open(out + '/pmc_b4096.json')), json.load(open(out + '/pmc_b65536.json'))
note = 'round 6 (tools/profile_round6.sh -> tools/pmc_percall.py): sums over the launches of ONE pairing call (two-program Miller loop, kernels running alone); FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md'
json.dump({'4096': dict(a, note=note), '65536': dict(b, note=note)}, open(out + '/hbm_traffic.json', 'w'), indent=1)
PY
python tools/verify_breakdown.py > $out/verify_breakdown_n65536.txt 2>&1
python tools/verify_breakdown.py 1 > $out/verify_breakdown_n1.txt 2>&1
bash tools/verify_timeline.sh 65536 r6_n65536 > /dev/null 2>&1; cp gpurun_out/verify_timeline_r6_n65536/timeline.txt $out/verify_timeline_n65536.txt
bash tools/verify_timeline.sh 1 r6_n1 > /dev/null 2>&1; cp gpurun_out/verify_timeline_r6_n1/timeline.txt $out/verify_timeline_n1.txt
( for rep in 1 2; do NBLS_POW_WIDE_MAX=0 python tools/latency_ab.py powers_one_lane_$rep 2>&1 | grep -a LATENCY_AB; python tools/latency_ab.py default_$rep 2>&1 | grep -a LATENCY_AB; done ) > $out/ab_latency.txt 2>&1
( NBLS_WIDE_MAX=0 python tools/wide_time.py lane_split; NBLS_WIDE_MAX=128 python tools/wide_time.py one_limb_per_lane ) 2>&1 | grep -a WIDE_TIME > $out/wide_time.txt
python tools/verify_sweep.py 65536 8 1,2,3 12,25 2>&1 | grep verifyBatch > $out/ab_verify.txt
python tools/ab_split_min.py 3072,4096,4608,6144,8191 2>&1 | grep SPLIT_AB > $out/ab_split_min.txt
python tools/sign_sizes.py 1,2048,8192,32768 > $out/sign_sizes.txt 2>&1
python tools/init_time.py > $out/init_time.txt 2>&1
( for rep in 1 2; do python tools/pair_ab.py default_$rep 2>&1 | grep PAIR_AB; done ) > $out/pair_ab.txt 2>&1
bash tools/clocks_under_load.sh > $out/clocks_under_load.txt 2>&1
python tools/stress_parity.py > $out/stress_parity.txt 2>&1
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err
rm -rf $out/b4096 $out/b65536 $out/b4096_inflight12 $out/pmc_b4096 $out/pmc_b65536     # the raw traces stay on the box; the summaries are merged back
ls -la $out
