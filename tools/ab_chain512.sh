B="python bench.py --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["batches_in_flight"])'
for rep in 1 2; do
  echo "rep=$rep 512 default:        $($B 2>/dev/null | python -c "$P")"
  echo "rep=$rep 512 NBLS_CHAIN_MAX=0: $(NBLS_CHAIN_MAX=0 $B 2>/dev/null | python -c "$P")"
  echo "rep=$rep 20 default d=12:   $($B --steps 20 --warmup 5 --inflight 12 2>/dev/null | python -c "$P")"
  echo "rep=$rep 20 nochain d=12:   $(NBLS_CHAIN_MAX=0 $B --steps 20 --warmup 5 --inflight 12 2>/dev/null | python -c "$P")"
  echo "rep=$rep 20 nochain d=20 fair0:   $(NBLS_FAIR=0 NBLS_CHAIN_MAX=0 $B --steps 20 --warmup 5 2>/dev/null | python -c "$P")"
done
