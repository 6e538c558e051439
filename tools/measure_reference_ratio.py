#!/usr/bin/env python3
"""Measures, in the build container (the only place the reference can run), the speed of the stand-in bench.py times on the GPU box (oracle/js_bigint_pairing.js)
relative to the REAL reference's pairing() under the same Node, alternating the two three times, and writes profiles/reference_ratio.json.  bench.py reads the
ratio from that file (cpu_baseline.js_bigint.ratio_to_reference) instead of carrying a literal.
Usage: python tools/strip_ts.py /root/reference /tmp/nbls_ref && python tools/measure_reference_ratio.py"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref_dir = sys.argv[1] if len(sys.argv) > 1 else '/tmp/nbls_ref'
secs = sys.argv[2] if len(sys.argv) > 2 else '5'
runs = []
for _ in range(3):
    r = json.loads(subprocess.check_output(['node', os.path.join(ROOT, 'tools', 'time_reference.mjs'), ref_dir, secs], text=True).strip().splitlines()[-1])
    s = json.loads(subprocess.check_output(['node', os.path.join(ROOT, 'oracle', 'js_bigint_pairing.js'), secs], text=True).strip().splitlines()[-1])
    assert r['ok'] and s['ok']
    runs.append({'reference_pairings_per_s': r['pairings_per_s'], 'standin_pairings_per_s': s['pairings_per_s'], 'ratio': round(s['pairings_per_s'] / r['pairings_per_s'], 4)})
out = {'what': 'oracle/js_bigint_pairing.js (stand-in timed on the GPU box) against the reference pairing(G1, G2) with fresh precomputes, same Node, build container, one core',
       'node': r['node'], 'date': time.strftime('%Y-%m-%d'), 'runs': runs, 'ratio': round(sum(x['ratio'] for x in runs) / len(runs), 4)}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'reference_ratio.json'), 'w'), indent=1)
print(json.dumps(out))
