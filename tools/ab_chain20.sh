# the driver's 20-step headline run with and without the chained final exponentiation (a chained wavefront runs 427 k instructions: coarse rounds at the end of a short run), and with
# the two-program Miller loop split points; three repeats each.  Usage (GPU box): bash tools/ab_chain20.sh > gpurun_out/ab_chain20.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["batches_in_flight"])'
for rep in 1 2 3; do
  echo "rep=$rep default:        $($B 2>/dev/null | python -c "$P")"
  echo "rep=$rep NBLS_CHAIN_MAX=0: $(NBLS_CHAIN_MAX=0 $B 2>/dev/null | python -c "$P")"
done
