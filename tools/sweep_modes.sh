# pairing batch: one-program (fused) against two-program Miller loop, with and without the two-halves split, per batch size
for n in 2048 4096 8192 12288 16384 32768 65536; do
  for mode in "NBLS_FUSED_MILLER=1 NBLS_HALVES_MIN=0" "NBLS_FUSED_MILLER=0 NBLS_HALVES_MIN=0" "NBLS_FUSED_MILLER=1 NBLS_HALVES_MIN=1" "NBLS_FUSED_MILLER=0 NBLS_HALVES_MIN=1"; do
    echo "$mode: $(env $mode python tools/exp_time.py $n 5 2>&1 | tail -1 | cut -c1-70)"
  done
done
