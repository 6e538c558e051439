#!/bin/bash
# Sweep lanes-per-item (W) of the Miller / exp-by-x step programs; prints pairings/s and per-program kernel time.
for b in 4096 65536; do
for mw in 12 16 20 32; do
for ew in 12 16 32; do
  out=$(NBLS_MILLER_W=$mw NBLS_EXPX_W=$ew timeout 120 python bench.py --steps 5 --warmup 1 --batch $b --no-cpu-baseline --verify-batch 0 --product-terms 0 2>/dev/null | tail -1)
  echo "batch=$b MILLER_W=$mw EXPX_W=$ew $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(round(d["value"]), "miller_ms", k["miller_fe"], "expx_ms", k["expx"])' 2>&1 | tail -1)"
done; done; done
