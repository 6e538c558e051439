#!/bin/bash
# round 5 A/B: the two-lane lane-split forms for calls of 1025 .. 2048 pairs (NBLS_LS2_MAX=0: the plain forms, as round 4) -- one call at a time, tools/pair_ab.py's call_2048_ms
export TMPDIR=/tmp
for rep in 1 2; do
  for v in 0 2048; do NBLS_LS2_MAX=$v python tools/pair_ab.py ls2max_$v 2>&1 | grep PAIR_AB | python -c "
import json, sys
for ln in sys.stdin:
    tag = ln.split()[1]; d = json.loads(ln.split(' ', 2)[2])
    print(tag, {k: d[k] for k in ('call_1024_ms', 'call_2048_ms', 'call_4096_ms')})"; done
done
