# in-flight depth of the headline leg at the driver's 20 steps: value per depth and hardware-queue count, three repeats each
for rep in 1 2 3; do for q in 16 32; do for d in 10 20; do
  echo "rep=$rep queues=$q inflight=$d steps=20: $(GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --inflight $d --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['batches_in_flight'])")"
done; done; done
