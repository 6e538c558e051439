#!/bin/bash
# round 5 A/B of the verifyBatch sub-batch form: Miller halves of the large sub-batch x its keys on a side stream, over sub-batch settings (tools/verify_sweep.py)
export TMPDIR=/tmp
for rep in 1 2; do
  for h in 0 1; do for k in 0 1; do
    NBLS_VERIFY_HALVES=$h NBLS_VERIFY_KEYS_SIDE=$k python tools/verify_sweep.py 65536 8 2,3 12,25,40 2>&1 | grep verifyBatch | sed "s/^/halves=$h keys_side=$k /"
  done; done
done
