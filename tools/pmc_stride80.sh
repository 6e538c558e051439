export TMPDIR=/tmp NBLS_HALVES_MIN=0 NBLS_SLOT_BYTES=80
out=$PWD/gpurun_out/prof_s80; rm -rf $out; mkdir -p $out
cmd="python bench.py --steps 3 --warmup 1 --batch 65536 --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0 --inflight 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1)); rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1
done
python tools/pmc_summary.py $out 4096
