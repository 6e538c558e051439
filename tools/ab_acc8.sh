# eight against four line tables per accumulator in the Miller product (verifyBatch of 65,536 signatures, product of 2^18 terms)
python -m pytest tests/test_gpu_adversarial.py -x -q -k "verify_batch_65536 or product" 2>&1 | tail -2
for m in 1000000000 32768; do
  echo "acc8_min=$m"
  NBLS_ACC8_MIN=$m python tools/verify_breakdown.py 65536 2>&1 | grep "verify_batch_dev\|lines_pq\|acc4\|acc8\|mul2"
  NBLS_ACC8_MIN=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sign-batch 0 --msm-points 0 --large-batch 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('verify ms', d['verify_batch']['ms'], 'in flight', d['verify_batch']['in_flight']['ms_per_call_amortised'], 'product ms', d['product']['ms_per_product'])"
done
