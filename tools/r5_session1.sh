#!/bin/bash
# round 5, GPU session 1: parity of the new pow kernels / pipeline / in-place tree, then A/B of the verifyBatch pipeline against round 4's two-phase form on ONE box
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=22
out=gpurun_out/r5s1; mkdir -p $out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $out/pytest.txt 2>&1
cat $out/pytest.txt | tail -5
( NBLS_VERIFY_PIPE=0 timeout 300 python tools/verify_sweep.py 65536 8 ) > $out/verify_old.txt 2>&1; tail -2 $out/verify_old.txt
( timeout 600 python tools/verify_sweep.py 65536 8 1,2,3,4,5,6,8 6,10,14,20 ) > $out/verify_sweep.txt 2>&1; grep verifyBatch $out/verify_sweep.txt
( NBLS_VERIFY_PIPE=0 timeout 300 python tools/verify_sweep.py 65536 8 ) > $out/verify_old2.txt 2>&1; tail -1 $out/verify_old2.txt
for n in 1 4096 16384; do
  ( NBLS_VERIFY_PIPE=0 timeout 200 python tools/verify_sweep.py $n 12 ) 2>&1 | tail -1
  ( timeout 200 python tools/verify_sweep.py $n 12 1,2,4 12 ) 2>&1 | grep verifyBatch
done > $out/verify_small.txt 2>&1; cat $out/verify_small.txt
( timeout 300 bash tools/verify_timeline.sh 65536 r5s1 ) > $out/timeline.log 2>&1; cp gpurun_out/verify_timeline_r5s1/timeline.txt $out/timeline_n65536.txt 2>/dev/null; head -3 $out/timeline_n65536.txt
