#!/usr/bin/env python3
"""Round 6: kernel time of the final exponentiation's programs for ONE element (the one-limb-per-lane interpreter against the lane-split forms): HIP-event durations"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module('noble-bls12-381_amd')
eng = pkg.Engine(0)
blob = bytes(range(1, 49)) * 12
for _ in range(5): eng.final_exp_batch(blob)
best = {}
for _ in range(20):
    eng.timing_enable(True); eng.final_exp_batch(blob); tm = eng.timing_read(); eng.timing_enable(False)
    for k, v in tm.items(): best[k] = min(best.get(k, 1e9), v[0])
print('WIDE_TIME', sys.argv[1] if len(sys.argv) > 1 else '', json.dumps({k: round(v, 4) for k, v in best.items()}), flush=True)
