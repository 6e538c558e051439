#!/bin/bash
# round 5, GPU session 2: verifyBatch as concurrent sub-batches: sweep, small sizes, timeline; new binding / fallback tests
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=22
out=gpurun_out/r5s2; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_binding.py tests/test_gpu_codec.py tests/test_gpu_adversarial.py -m gpu -x -q 2>&1 | tail -15 ) > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
( NBLS_VERIFY_PIPE=0 timeout 300 python tools/verify_sweep.py 65536 8 ) > $out/verify_old.txt 2>&1; tail -1 $out/verify_old.txt
( timeout 900 python tools/verify_sweep.py 65536 8 1,2,3,4,5,6,8 8,12,17,25 ) > $out/verify_sweep.txt 2>&1; grep verifyBatch $out/verify_sweep.txt
( NBLS_VERIFY_PIPE=0 timeout 300 python tools/verify_sweep.py 65536 8 ) > $out/verify_old2.txt 2>&1; tail -1 $out/verify_old2.txt
for n in 8192 16384 32768; do
  ( NBLS_VERIFY_PIPE=0 timeout 200 python tools/verify_sweep.py $n 12 ) 2>&1 | tail -1
  ( timeout 200 python tools/verify_sweep.py $n 12 1,2,3,4 12,25 ) 2>&1 | grep verifyBatch
done > $out/verify_small.txt 2>&1; cat $out/verify_small.txt
( timeout 300 bash tools/verify_timeline.sh 65536 r5s2 ) > $out/timeline.log 2>&1; cp gpurun_out/verify_timeline_r5s2/timeline.txt $out/timeline_n65536.txt 2>/dev/null; head -3 $out/timeline_n65536.txt
