#!/bin/bash
# A/B of the lane-split (latency) forms: NBLS_LS_MAX=0 (throughput forms at every size) against NBLS_LS_MAX=1024, pairing batches of 1 .. 1024; then the pairing
# parity tests with the lane-split forms switched on.  Usage (GPU box): tools/ab_ls.sh > gpurun_out/ab_ls.txt
cd "$(dirname "$0")/.."
for n in 1 16 64 256 512 1024; do
  for m in 0 1024; do echo -n "LS_MAX=$m "; NBLS_LS_MAX=$m python tools/exp_time.py $n 20 2>&1 | tail -1; done
done
echo "== parity, NBLS_LS_MAX=1024"
NBLS_LS_MAX=1024 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_sign.py tests/test_gpu_reference_vectors.py -x -q 2>&1 | tail -5
