#!/bin/bash
# round 5, GPU session 3: slot placement A/B (NBLS_LDS_LAYOUT=0 / 1, interleaved twice), parity of the placed programs, verifyBatch with stream priorities
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=22
out=gpurun_out/r5s3; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_adversarial.py tests/test_gpu_prepared.py tests/test_gpu_tower.py -m gpu -x -q 2>&1 | tail -6 ) > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for rep in 1 2; do
  NBLS_LDS_LAYOUT=0 timeout 300 python tools/pair_ab.py compiled_$rep 2>&1 | grep PAIR_AB
  NBLS_LDS_LAYOUT=1 timeout 300 python tools/pair_ab.py placed_$rep 2>&1 | grep PAIR_AB
done > $out/pair_ab.txt 2>&1; cat $out/pair_ab.txt
for p in 0 1; do
  ( NBLS_VERIFY_PRIO=$p timeout 300 python tools/verify_sweep.py 65536 8 2,3,4 12,25,40 ) 2>&1 | grep verifyBatch | sed "s/^/prio=$p /"
done > $out/verify_prio.txt 2>&1; cat $out/verify_prio.txt
( NBLS_LDS_LAYOUT=0 timeout 300 python tools/verify_sweep.py 65536 8 2 12 ) 2>&1 | grep verifyBatch | sed "s/^/compiled placement: /" | tee -a $out/verify_prio.txt
