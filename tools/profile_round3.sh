#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun; about five minutes).  Everything lands in gpurun_out/prof_r3/; the summaries that are kept are
# copied to profiles/round3_* by tools/collect_round3.sh.  Counters are collected in their own passes with --kernel-trace only.
#   1. rocprofv3 --kernel-trace --stats of the default bench configuration, one 4096-pairing call at a time     -> kernel_stats_b4096.csv
#   2. the same with twelve calls in flight (the configuration of `value`), plus the SPAN of the VM kernels of the timed region (first start -> last
#      end) next to the step count, so that frac_at_value can be recomputed from the trace                      -> kernel_stats_b4096_inflight12.csv, inflight12_span.json
#   3. the same for one 65,536-pairing call at a time                                                          -> kernel_stats_b65536.csv
#   4. PMC passes at 4096 and 65,536 (tools/pmc_summary.py)                                                    -> pmc_b4096.csv, pmc_b65536.csv
#   5. per-kernel times of one verifyBatch of 65,536 signatures and of one verify (tools/verify_breakdown.py)   -> verify_breakdown*.txt
#   6. the JSON line of a default bench run and of the driver's arguments                                      -> bench_default.json, bench_driver_args.json
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_r3; rm -rf $out; mkdir -p $out
common="--no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
stats() {   # <name> <bench args>
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1 -- python bench.py $2 $common > $out/$1.log 2>&1
  f=$(find $out/$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_$1.csv
}
export NBLS_HALVES_MIN=0      # kernel traces and counters of kernels running alone: a large call is not split into two overlapping halves here
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
marks = [i for i, r in enumerate(rows) if 'fill' in r['Kernel_Name'].lower()]     # bench.py --mark-timed-region: one fill kernel before and one after the timed steps
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
export NBLS_HALVES_MIN=0
for b in 4096 65536; do
  cmd="python bench.py --steps 3 --warmup 1 --batch $b --inflight 1 $common"
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_b$b/pmc$i -- $cmd > $out/pmc_b${b}_$i.log 2>&1
  done
  if [ $b = 4096 ]; then NBLS_FUSED_MILLER=1 python tools/pmc_summary.py $out/pmc_b$b 4096 > $out/pmc_b$b.csv; else python tools/pmc_summary.py $out/pmc_b$b 4096 > $out/pmc_b$b.csv; fi
done
unset NBLS_HALVES_MIN
python tools/hbm_traffic.py $out/pmc_b4096.csv $out/pmc_b65536.csv > $out/hbm_traffic.json
python tools/verify_breakdown.py > $out/verify_breakdown_n65536.txt 2>&1
python tools/verify_breakdown.py 1 > $out/verify_breakdown_n1.txt 2>&1
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err
/opt/rocm/lib/llvm/bin/llvm-readelf --notes noble-bls12-381_amd/libnbls.so 2>/dev/null | grep -E "\.name:|\.vgpr_count|\.sgpr_count|spill_count|\.group_segment_fixed_size" > $out/kernel_resources.txt
ls -la $out/*.csv $out/*.txt $out/*.json
