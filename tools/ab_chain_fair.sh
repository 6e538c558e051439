B="python bench.py --no-cpu-baseline --verify-batch 0 --product-terms 0 --sign-batch 0 --msm-points 0 --large-batch 0"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["batches_in_flight"])'
for rep in 1 2 3 4; do
  echo "rep=$rep 20 chain fair:     $($B --steps 20 --warmup 5 2>/dev/null | python -c "$P")"
  echo "rep=$rep 20 chain fair0:    $(NBLS_FAIR=0 $B --steps 20 --warmup 5 2>/dev/null | python -c "$P")"
  echo "rep=$rep 20 nochain fair:   $(NBLS_CHAIN_MAX=0 $B --steps 20 --warmup 5 2>/dev/null | python -c "$P")"
  echo "rep=$rep 20 nochain fair0:  $(NBLS_FAIR=0 NBLS_CHAIN_MAX=0 $B --steps 20 --warmup 5 2>/dev/null | python -c "$P")"
done
for rep in 1 2; do
  echo "rep=$rep 512 nochain fair:  $(NBLS_CHAIN_MAX=0 $B 2>/dev/null | python -c "$P")"
  echo "rep=$rep 512 nochain fair0: $(NBLS_FAIR=0 NBLS_CHAIN_MAX=0 $B 2>/dev/null | python -c "$P")"
done
