#!/bin/bash
# Two PMC passes + one kernel trace over the pairing pipeline at one batch size (a lighter tools/profile_round.sh). Usage: tools/pmc_quick.sh <tag> [batch]
tag=${1:-q}; batch=${2:-65536}
export TMPDIR=/tmp
export NBLS_HALVES_MIN=0
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
cmd="python bench.py --steps 3 --warmup 1 --batch $batch --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0 --inflight 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1
done
python tools/pmc_summary.py $out 4096 > gpurun_out/pmc_${tag}_b$batch.csv
cat gpurun_out/pmc_${tag}_b$batch.csv
