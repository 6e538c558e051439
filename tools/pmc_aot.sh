#!/bin/bash
# PMC passes over the pairing pipeline (two-program Miller loop, one call at a time) with the ahead-of-time kernels on, summarised per kernel name.
# Usage: tools/pmc_aot.sh <tag> [batch]
tag=${1:-aot}; batch=${2:-65536}
export TMPDIR=/tmp NBLS_HALVES_MIN=0 NBLS_FUSED_MILLER=0
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
cmd="python tools/exp_time.py $batch 2"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1
done
python tools/pmc_byname.py $out 4096 > gpurun_out/pmc_${tag}_b$batch.csv
cat gpurun_out/pmc_${tag}_b$batch.csv
