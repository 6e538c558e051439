#!/bin/bash
# round 5, GPU session 5: fixed-base getPublicKey and the psi-split sign ladder (parity + A/B), slot placements of the point programs on verifyBatch
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=22
out=gpurun_out/r5s5; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_sign.py tests/test_gpu_codec.py tests/test_gpu_binding.py tests/test_debug_build.py -m gpu -x -q 2>&1 | tail -8 ) > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for v in "NBLS_G1_FIXED=0 NBLS_G2_GLS=0" "NBLS_G1_FIXED=1 NBLS_G2_GLS=1" "NBLS_G1_FIXED=0 NBLS_G2_GLS=0" "NBLS_G1_FIXED=1 NBLS_G2_GLS=1"; do
  env $v timeout 300 python tools/mul_time.py 2>&1 | tail -4 | sed "s/^/$v: /"
done > $out/mul_ab.txt 2>&1; cat $out/mul_ab.txt
for l in 0 1 0 1; do ( NBLS_LDS_LAYOUT=$l timeout 300 python tools/verify_sweep.py 65536 8 2 12 ) 2>&1 | grep verifyBatch | sed "s/^/layout=$l /"; done > $out/verify_layout.txt 2>&1; cat $out/verify_layout.txt
