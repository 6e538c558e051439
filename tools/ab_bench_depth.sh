#!/bin/bash
# the driver's command line, interleaved A/B of an environment switch of bench.py (first use: 10 / 12 / 14 / 20 pool contexts -- no difference; now: oracle reference before / after the GPU parity calls), secondary legs off: `value` as bench.py measures it (ONE burst of 20 calls after 5 warm-up calls, right after an idle phase)
export TMPDIR=/tmp
off="--no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0 --config3-pairings 0"
for rep in 1 2 3 4 5 6; do for d in 0 96; do
  NBLS_BENCH_PREWARM=$d python bench.py --gpus 1 --steps 20 --warmup 5 $off 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prewarm', '$d', 'inflight', d['config']['batches_in_flight'], 'value %.4f M' % (d['value'] / 1e6), 'single', d['single_call']['ms_per_batch'])"
done; done
