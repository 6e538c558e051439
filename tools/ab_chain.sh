# A/B of the chained final exponentiation (one launch for EXPX, FE_MID1, EXPX x 3, FE_MID2, EXPX) against one launch per program
python -m pytest tests/test_gpu_pairing.py -x -q 2>&1 | tail -2
for ch in 0 1; do
  for n in 1 1024 4096 16384 65536; do
    echo "chain=$ch"; NBLS_CHAIN=$ch python tools/exp_time.py $n 5 2>&1 | tail -1
  done
done
