#!/bin/bash
# Round-2 profile set (run on the GPU box through gpurun; about four minutes).  Everything lands in gpurun_out/prof_r2/ and the summaries that
# are kept are copied to profiles/ by hand (tools/README.md).  Counters are collected in their own passes with --kernel-trace only.
#   1. rocprofv3 --kernel-trace --stats of the default bench configuration, one 4096-pairing call at a time     -> kernel_stats_b4096.csv
#   2. the same with twelve calls in flight (the configuration of `value`)                                     -> kernel_stats_b4096_inflight12.csv
#   3. the same for one 65,536-pairing call at a time                                                          -> kernel_stats_b65536.csv
#   4. PMC passes at 4096 and 65,536 (tools/pmc_summary.py)                                                    -> pmc_b4096.csv, pmc_b65536.csv
#   5. per-kernel times of one verifyBatch of 65,536 signatures (tools/verify_breakdown.py)                     -> verify_breakdown.txt
#   6. the JSON line of a default bench run                                                                    -> bench_default.json
export TMPDIR=/tmp
export NBLS_HALVES_MIN=0      # kernel traces and counters of kernels running alone: a large call is not split into two overlapping halves here
out=$PWD/gpurun_out/prof_r2; mkdir -p $out
common="--no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
stats() {   # <name> <bench args>
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1 -- python bench.py $2 $common > $out/$1.log 2>&1
  f=$(find $out/$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_$1.csv
}
stats b4096 "--steps 16 --warmup 2 --batch 4096 --inflight 1"
stats b4096_inflight12 "--steps 192 --warmup 12 --batch 4096 --inflight 12"
stats b65536 "--steps 3 --warmup 1 --batch 65536 --inflight 1"
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
python tools/verify_breakdown.py > $out/verify_breakdown.txt 2>&1
python tools/verify_breakdown.py 1 > $out/verify_breakdown_n1.txt 2>&1
python bench.py > $out/bench_default.json 2> $out/bench_default.err
ls -la $out/*.csv $out/*.txt $out/*.json
