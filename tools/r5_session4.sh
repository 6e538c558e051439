#!/bin/bash
# round 5, GPU session 4: the whole GPU suite on the current tree, then bench.py with the driver's arguments (new legs: config3, pool, roofline.hbm)
export TMPDIR=/tmp
out=gpurun_out/r5s4; mkdir -p $out
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
( timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err ); echo "bench rc=$?"; tail -3 $out/bench_driver_args.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5s4/bench_driver_args.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'single', d['single_call'], )
print('roofline', {k: r[k] for k in ('kernel', 'aot_programs', 'miller_form', 'final_exp_middle', 'achieved', 'frac', 'frac_at_value', 'traffic', 'hbm')})
print('large', {k: v for k, v in r['large_batch'].items() if k != 'kernel_ms'})
print('verify', {k: v for k, v in d['verify_batch'].items() if k in ('value', 'ms', 'in_flight', 'single_verify_ms', 'host_call_ms')})
print('config3', d['config3']); print('pool', d['pool']); print('product', d['product']); print('sign', d['sign']); print('msm', d['msm'])
PY
