export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_lat; mkdir -p $out
cmd="python bench.py --steps 3 --warmup 1 --batch 4096 --no-cpu-baseline --verify-batch 0 --product-terms 0"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $out/sq_counters.txt
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pmc1 -- $cmd > $out/pmc1.log 2>&1
tail -3 $out/pmc1.log
