#!/bin/bash
# LDS-side counters of the pairing pipeline for a given engine library. Usage: tools/prof_lds.sh <tag> <library> [batch]
export TMPDIR=/tmp
tag=$1; lib=$2; batch=${3:-4096}
out=$PWD/gpurun_out/prof_lds_$tag; mkdir -p $out
NBLS_LIBRARY=$lib rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pmc1 -- python tools/exp_time.py $batch 3 > $out/pmc1.log 2>&1
NBLS_LIBRARY=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc2 -- python tools/exp_time.py $batch 3 > $out/pmc2.log 2>&1
