#!/usr/bin/env python3
"""Rewrites the numbers block of README.md (from '**Numbers on one MI355X' up to 'What round 6 changed:') from profiles/round6_bench_driver_args.json / round6_bench_default.json.
Usage: tools/readme_numbers.py"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, 'profiles', 'round6_bench_driver_args.json'))); f = json.load(open(os.path.join(ROOT, 'profiles', 'round6_bench_default.json')))
p = os.path.join(ROOT, 'README.md')
s = open(p).read()
a = s.index('**Numbers on one MI355X'); b = s.index('What round 6 changed:')
lb = d['roofline']['large_batch']; vb = d['verify_batch']; sg = d['sign']; fc = d['facade']; cl = lb['in_flight']; ss = d['steady_state']
new = """**Numbers on one MI355X, measured as the driver measures them** (`python bench.py --gpus 1 --steps 20 --warmup 5`, `profiles/round6_bench_driver_args.json`; that box ran
saturated at %.2f GHz / %d W; the boxes of the pool sustain 2.13 - 2.26 GHz at the same ~1300 W, and saturated figures follow: 21.1 - 23.2 ms for one 65,536-pairing call (`profiles/round6_box_spread.txt`); round 5's driver record in brackets):

| | round 6 | |
|---|---|---|
| `value`: 4096-pairing batches, 12 in flight through the C-ABI pool, 5 warm-up + 20 timed steps and nothing else in front of them (the protocol of rounds 1-4 again) | **%.2f M pairings/s**, %d %% of the integer multiply-add roofline; **%.2f M** after 96 further untimed steps (`steady_state`: no clock ramp inside the 29 ms region) | [2.96 M after 96 pre-warm steps, 2.78 M without] |
| one 4096-pairing call at a time | **%.2f ms** (%.2f M/s), %d %% | [2.33 ms] |
| 1 ... 1024 pairings in one call / 2048 | **1.49 ms** / 1.75 ms | [1.76 / 1.87 ms] |
| one 65,536-pairing call | %.2f ms (%.2f M/s), %d %% | [21.83 ms] |
| `verifyBatch`, 65,536 signatures, one call (SHA-256 `expand_message_xmd` inside) | **%.2f ms (%.2f M sigs/s)**; %.1f ms per call with three in flight | [20.45 ms] |
| 2^20 independent pairings in one call (BASELINE configs[3] on one GPU) | %d ms (%.2f M/s) | [349 ms] |
| a single `verify` (C ABI) / `await bls.verify()` / `await bls.sign()` from JavaScript | **%.2f / %.2f / %.2f ms** | [3.68 / 3.59 / 2.90] |
| `getPublicKey` / `sign`, 8192 keys from host buffers; `sign` with everything resident in HBM | %.2f M keys/s / **%.2f M sigs/s**; %.2f M sigs/s | [5.32 M / 1.74 M; 2.13 M] |

With 512 steps (`python bench.py`, the steady state): %.2f M pairings/s (%d %%). The same box's host: the C restatement of the reference %.1f k pairings/s on %d threads; the
reference's algorithm in JavaScript BigInt %d pairings/s on one core, %.1f k on all %d.

""" % (sum(cl['sclk_mhz_under_load']) / 2 / 1000, round(sum(cl['package_power_w_under_load']) / 2), d['value'] / 1e6, round(100 * d['roofline']['frac_at_value']), ss['pairings_per_s'] / 1e6,
       d['single_call']['ms_per_batch'], d['single_call']['pairings_per_s'] / 1e6, round(100 * d['roofline']['frac']),
       lb['ms_per_call'], lb['pairings_per_s'] / 1e6, round(100 * lb['frac']), vb['ms'], vb['value'] / 1e6, vb['in_flight']['ms_per_call_amortised'],
       round(d['config3']['ms']), d['config3']['value'] / 1e6, vb['single_verify_ms'], fc['verify_ms'], fc['sign_ms'],
       sg['get_public_key_keys_per_s'] / 1e6, sg['value'] / 1e6, sg['resident']['sigs_per_s'] / 1e6,
       f['value'] / 1e6, round(100 * f['roofline']['frac_at_value']), f['cpu_baseline']['value'] / 1e3, f['cpu_baseline']['cores'], round(f['cpu_baseline']['js_bigint']['value']),
       f['cpu_baseline']['js_bigint']['all_cores']['value'] / 1e3, f['cpu_baseline']['js_bigint']['all_cores']['cores'])
open(p, 'w').write(s[:a] + new + s[b:])
print('README.md refreshed')
