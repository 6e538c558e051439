# the headline leg (512 steps, twelve contexts) under hardware-queue limits and with / without the chained final exponentiation in the in-flight contexts; then verifyBatch alone
# in a fresh process (default queue limit) and inside a bench run.  Usage (GPU box): bash tools/ab_pipeline.sh > gpurun_out/ab_pipeline.txt
B="python bench.py --no-cpu-baseline --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d.get("verify_batch") or {}; print(d["value"], d["ms_per_step"], d["config"]["batches_in_flight"], "single", d["single_call"]["ms_per_batch"], "verify", v.get("ms"), v.get("single_verify_ms"))'
for rep in 1 2; do
  for q in 16 22 32; do
    echo "rep=$rep queues=$q unchained: $(GPU_MAX_HW_QUEUES=$q $B --verify-batch 0 2>/dev/null | python -c "$P")"
    echo "rep=$rep queues=$q chained:   $(NBLS_PIPELINE_CHAIN=1 GPU_MAX_HW_QUEUES=$q $B --verify-batch 0 2>/dev/null | python -c "$P")"
  done
done
python tools/verify_time.py 65536 10 2>&1 | tail -1
for q in 8 16 22; do echo "bench verify leg, queues=$q: $(GPU_MAX_HW_QUEUES=$q $B 2>/dev/null | python -c "$P")"; done
python tools/verify_time.py 65536 10 2>&1 | tail -1
