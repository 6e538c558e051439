#!/bin/bash
# round 6 A/B: the two-lane forms at configs[1]'s own size.  NBLS_LS2_MAX in {2048, 3072, 4096}, with the fused Miller program forced (its LS2 form exists) or the
# default LINES + ACC split (only EXPX has an LS2 form there); one call at a time and the in-flight rate (tools/pair_ab.py), interleaved, twice
export TMPDIR=/tmp
for rep in 1 2; do
  for cfg in "2048 -1" "3072 1" "4096 -1" "4096 1"; do
    set -- $cfg
    NBLS_LS2_MAX=$1 NBLS_FUSED_MILLER=$2 python tools/pair_ab.py ls2max_$1_fused_$2 2>&1 | grep PAIR_AB | python -c "
import json, sys
for ln in sys.stdin:
    tag = ln.split()[1]; d = json.loads(ln.split(' ', 2)[2])
    print(tag, {k: d[k] for k in d if k != 'smi_under_load'}, d.get('smi_under_load'))"; done
done
