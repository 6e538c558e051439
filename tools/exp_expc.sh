#!/bin/bash
# A/B of the compressed cyclotomic squarings (NBLS_EXPC_MIN: items from which expx() in csrc/pipelines_pairing.cpp uses them; a huge value = never)
mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_pairing.py -q -x -k "compressed or final_exp or kilic or batch_4096 or 131072" > gpurun_out/r3e/pytest.log 2>&1; tail -3 gpurun_out/r3e/pytest.log
for m in 1000000000 0; do
  export NBLS_EXPC_MIN=$m; echo "NBLS_EXPC_MIN=$m"
  python tools/exp_time.py 4096 20 2>&1 | tail -1
  python tools/exp_time.py 16384 10 2>&1 | tail -1
  python tools/exp_time.py 65536 5 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r3e/expc_ab.txt
unset NBLS_EXPC_MIN
python bench.py --steps 256 --warmup 16 --verify-batch 0 --msm-points 0 --sign-batch 0 --product-terms 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', d['value'], 'single', d['single_call']['ms_per_batch'], 'large', d['roofline']['large_batch']['ms_per_call'], d['roofline']['large_batch']['in_flight'])" | tee -a gpurun_out/r3e/expc_ab.txt
