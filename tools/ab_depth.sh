# in-flight depth of the headline leg at the driver's 20 steps and at 512: value per depth (hardware queues 16 / 24 / 32)
for q in 16 32; do for d in 10 12 16 20; do for st in 20 512; do
  echo "queues=$q inflight=$d steps=$st: $(GPU_MAX_HW_QUEUES=$q python bench.py --steps $st --warmup 5 --inflight $d --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['batches_in_flight'], d['single_call']['ms_per_batch'])")"
done; done; done
