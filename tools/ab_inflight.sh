# in-flight leg of bench.py (twelve 4096-pairing calls overlapping) with and without the chained final exponentiation
for ch in 0 1; do
  echo "chain=$ch"
  NBLS_CHAIN=$ch python bench.py --steps 512 --warmup 16 --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['single_call']['ms_per_batch'], d['config']['batches_in_flight'])"
done
