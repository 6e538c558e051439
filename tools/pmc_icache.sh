#!/bin/bash
# Instruction-supply counters of the pairing programs at 4096 (one wavefront per SIMD) and 65,536 pairings: instruction-cache requests / hits / misses, instruction fetches
# and their mean latency (SQ_IFETCH_LEVEL / SQ_IFETCH), branches.   Output: gpurun_out/pmc_icache/summary_b*.csv
export TMPDIR=/tmp NBLS_HALVES_MIN=0
out=$PWD/gpurun_out/pmc_icache; rm -rf $out; mkdir -p $out
common="--no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
for b in 4096 65536; do
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_b$b/pmc$i -- python bench.py --steps 3 --warmup 1 --batch $b --inflight 1 $common > $out/pmc_b${b}_$i.log 2>&1
  done
  if [ $b = 4096 ]; then NBLS_FUSED_MILLER=1 python tools/pmc_summary.py $out/pmc_b$b 4096 > $out/summary_b$b.csv; else python tools/pmc_summary.py $out/pmc_b$b 4096 > $out/summary_b$b.csv; fi
  cat $out/summary_b$b.csv
done
