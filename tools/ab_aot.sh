set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pairing.py -x -q 2>&1 | tail -5
for aot in 0 1; do
  for n in 4096 65536; do
    NBLS_AOT=$aot NBLS_HALVES_MIN=0 python tools/exp_time.py $n 5 2>&1 | tail -1
    NBLS_AOT=$aot NBLS_HALVES_MIN=0 NBLS_FUSED_MILLER=0 python tools/exp_time.py $n 5 2>&1 | tail -1
  done
done
