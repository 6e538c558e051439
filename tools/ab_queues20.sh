# the driver's command line (20 steps -> 20 in-flight contexts) under different hardware-queue limits: the secondary legs (verifyBatch, sign, MSM) create further streams, and above ~24
# user queues the process oversubscribes the hardware queue slots.  Usage (GPU box): bash tools/ab_queues20.sh > gpurun_out/ab_queues20.txt
P='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j["value"], j["single_call"]["ms_per_batch"], "verify", j["verify_batch"]["ms"], j["verify_batch"]["single_verify_ms"], "sign", j["sign"]["ms"], "msm", j["msm"]["ms"], "agg65536", j["aggregate"]["aggregate_public_keys_65536_ms"])'
for q in ${QS:-32 24 20 16}; do
  echo "GPU_MAX_HW_QUEUES=$q: $(GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline --product-terms 0 2>/dev/null | python -c "$P")"
done
