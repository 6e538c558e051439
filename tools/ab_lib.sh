#!/bin/bash
# A/B of two builds of the engine on one box: noble-bls12-381_amd/variants/libnbls_pre.so (NBLS_LIBRARY) against the in-tree library, tools/pair_ab.py, interleaved
export TMPDIR=/tmp
for rep in 1 2 3; do
  NBLS_LIBRARY=$PWD/noble-bls12-381_amd/variants/libnbls_pre.so python tools/pair_ab.py pre_$rep 2>&1 | grep PAIR_AB | cut -c1-420
  python tools/pair_ab.py new_$rep 2>&1 | grep PAIR_AB | cut -c1-420
done
