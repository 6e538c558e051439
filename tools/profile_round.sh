#!/bin/bash
# rocprofv3 passes over one bench invocation (pairing batch only). Usage: tools/profile_round.sh <tag> [batch]
# Writes gpurun_out/prof_<tag>/{trace,pmc1..5}/ ; counters are collected in their own passes (no tracing domains besides kernel-trace).
tag=${1:-r1}; batch=${2:-4096}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
cmd="python bench.py --steps 3 --warmup 1 --batch $batch --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --inflight 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $cmd > $out/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1
done
find $out -name "*.csv" | head -30
