#!/bin/bash
# A/B of the SDST rotation (csrc/vm_exec.h NBLS_SDST_PAIRS): variants built by tools/exp_variants.sh sdstN "-DNBLS_SDST_PAIRS=N"
mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_pairing.py tests/test_gpu_adversarial.py -q -x > gpurun_out/r3c/pytest_default.log 2>&1; tail -2 gpurun_out/r3c/pytest_default.log
for v in "" sdst0 sdst2 sdst8; do
  if [ -n "$v" ]; then export NBLS_LIBRARY=$PWD/noble-bls12-381_amd/variants/libnbls_$v.so; else unset NBLS_LIBRARY; fi
  python tools/exp_time.py 4096 20 2>&1 | tail -1
  python tools/exp_time.py 65536 5 2>&1 | tail -1
  python bench.py --steps 256 --warmup 16 --verify-batch 0 --msm-points 0 --sign-batch 0 --large-batch 0 --product-terms 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', d['value'], 'single', d['single_call']['ms_per_batch'])"
done 2>&1 | tee gpurun_out/r3c/sdst_ab.txt
