# in-flight depth x GPU_MAX_HW_QUEUES sweep of the headline leg (bench.py --inflight D); usage: bash tools/exp_queues.sh
A="--no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
for q in 16 24 32; do for d in 10 12 14 16 20; do
echo -n "queues $q depth $d steps 560: "; GPU_MAX_HW_QUEUES=$q timeout 120 python bench.py --steps 560 --warmup 40 --inflight $d $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']))"
done; done
for d in 7 10 12 14 20; do
echo -n "queues 16 depth $d steps 20 warmup 5: "; GPU_MAX_HW_QUEUES=16 timeout 120 python bench.py --steps 20 --warmup 5 --inflight $d $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']))"
done
