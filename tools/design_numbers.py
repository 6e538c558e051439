#!/usr/bin/env python3
"""Markdown rows of DESIGN.md section 6 from the committed bench JSON lines (profiles/roundN_bench_default.json, ..._bench_driver_args.json).  Usage: tools/design_numbers.py [round]"""
import json, sys
r = sys.argv[1] if len(sys.argv) > 1 else '6'
d = json.load(open('profiles/round%s_bench_default.json' % r)); a = json.load(open('profiles/round%s_bench_driver_args.json' % r))
v = d['verify_batch']; f = d.get('facade') or {}; c = d['cpu_baseline']; j = c['js_bigint']
print('| `value`: 4096-pairing batches, %d in flight (%d steps) | **%.3f M pairings/s** (%.3f ms per batch) | %.3f (`roofline.frac_at_value`) |' % (d['config']['batches_in_flight'], d['steps'], d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac_at_value']))
ss = a.get('steady_state') or {}
print('| the same at the driver\'s arguments: %d warm-up + %d timed steps, nothing in front of them (%d contexts; a %.0f ms region on a chip that was idle a moment before) | %.3f M pairings/s (`value` of that line; %.3f M after %d further untimed steps: `steady_state`) | %.3f |' % (a['warmup'], a['steps'], a['config']['batches_in_flight'], a['ms_per_step'] * a['steps'], a['value'] / 1e6, (ss.get('pairings_per_s') or 0) / 1e6, ss.get('prewarm_steps', 0), a['roofline']['frac_at_value']))
print('| `single_call`: one 4096-pairing call at a time | %.3f ms (%.2f M pairings/s) | **%.4f** (top-level `roofline.frac`) |' % (d['single_call']['ms_per_batch'], d['single_call']['pairings_per_s'] / 1e6, d['roofline']['frac']))
lb = d['roofline']['large_batch']
print('| `roofline.large_batch`: one 65,536-pairing call | %.2f ms (%.2f M pairings/s) | %.3f; three calls in flight %.3f |' % (lb['ms_per_call'], lb['pairings_per_s'] / 1e6, lb['frac'], lb['in_flight']['roofline_frac']))
print('| `verify_batch`: 65,536 signatures, one call at a time, messages / keys / signature resident in HBM, `expand_message_xmd` inside | %.2f ms (%.2f M sigs/s) | %.3f on the reference\'s count, **%.3f on the executed algorithm**; three calls in flight %.2f ms per call (%.2f M sigs/s) |' % (v['ms'], v['value'] / 1e6, v['roofline']['frac'], v['roofline']['frac_executed'], v['in_flight']['ms_per_call_amortised'], v['in_flight']['sigs_per_s'] / 1e6))
print('| one `verify` from host buffers (C ABI) / `await bls.verify(...)` from JavaScript / `await bls.sign(...)` | %.2f ms / %s ms / %s ms | critical path, section 4 |' % (v['single_verify_ms'], f.get('verify_ms'), f.get('sign_ms')))
print('| `product`: 2^18-term Miller product + one final exponentiation | %.1f ms (%.2f M terms/s) | |' % (d['product']['ms_per_product'], d['product']['value'] / 1e6))
print('| `sign` 8192 keys from host buffers / resident in HBM (`nbls_sign_batch_dev`) / `getPublicKey` | %.2f M / %s M sigs/s / %.2f M keys/s | |' % (d['sign']['value'] / 1e6, ('%.2f' % (d['sign']['resident']['sigs_per_s'] / 1e6)) if d['sign'].get('resident') else '–', d['sign']['get_public_key_keys_per_s'] / 1e6))
if d.get('config3'): print('| `config3`: BASELINE configs[3], %d independent pairings in ONE call per rank (1 rank) | %.1f ms (%.2f M pairings/s) | %.3f |' % (d['config3']['pairings'], d['config3']['ms'], d['config3']['value'] / 1e6, d['config3']['roofline_frac_per_gpu']))
print('| MSM G1, 65,536 points, 255-bit scalars | %.1f M points/s | |' % (d['msm']['value'] / 1e6))
print('| hash-to-G2 / hash-to-G1, 16,384 messages from host buffers | %.2f / %.2f M msgs/s | |' % (d['aggregate']['hash_to_g2_msgs_per_s'] / 1e6, d['aggregate']['hash_to_g1_msgs_per_s'] / 1e6))
print()
print('CPU: C oracle %.0f pairings/s on %d threads; JS BigInt one core %.1f, all %d cores %.0f pairings/s; ratio to the reference %.3f -> reference estimate %.1f / %.0f' % (c['value'], c['cores'], j['value'], j['all_cores']['cores'], j['all_cores']['value'], j['ratio_to_reference'], j['reference_estimate']['one_core'], j['reference_estimate']['all_cores']))
