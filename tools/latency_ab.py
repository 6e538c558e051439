#!/usr/bin/env python3
"""Round 6: the latency legs in a few seconds -- one verify, one sign, one hash-to-G2, one signature / key decompression through the C ABI (host buffers), medians of 200 calls;
with per-kernel HIP-event times of one verify.  Usage: [NBLS_POW_WIDE_MAX=0] tools/latency_ab.py [tag]"""
import hashlib, importlib, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: F401
import oracle_py
pkg = importlib.import_module('noble-bls12-381_amd')
tag = sys.argv[1] if len(sys.argv) > 1 else ''
o = oracle_py.load(rebuild=False); eng = pkg.Engine(0)
sk = (int.from_bytes(hashlib.sha256(b'lat-sk').digest(), 'big') % (1 << 254) + 1).to_bytes(32, 'big')
msg = hashlib.sha256(b'lat-msg').digest(); pk = o.get_public_key(sk); sig = o.sign(msg, sk)[1]
assert eng.verify_batch(sig, [msg], [pk]) is True and eng.verify_batch(sig, [msg + b'x'], [pk]) is False
assert eng.sign_batch([msg], [sk]) == [sig]


def med(f, reps=200):
    for _ in range(10): f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(statistics.median(ts), 4), round(min(ts), 4)


res = {'verify_ms': med(lambda: eng.verify_batch(sig, [msg], [pk]))}
res['sign_ms'] = med(lambda: eng.sign_batch_affine([msg], [sk]))
res['hash_to_g2_ms'] = med(lambda: eng.hash_to_g2_batch([msg]))
res['g2_decompress_ms'] = med(lambda: eng.decompress_batch(sig, g2=True))
res['g1_decompress_ms'] = med(lambda: eng.decompress_batch(pk, g2=False))
try:
    g1a = eng.decompress_batch(pk, g2=False)[0]; g2a = eng.decompress_batch(sig, g2=True)[0]
    res['pairing1_ms'] = med(lambda: eng.pairing_batch(g1a, g2a))
    res['pairing1_validated_ms'] = med(lambda: eng.pairing_batch(g1a, g2a, validate=True))
    res['miller_product2_validated_ms'] = med(lambda: eng.miller_product(g1a * 2, g2a * 2, validate=True))
    assert eng.pairing_batch(g1a, g2a, validate=True)[0] == eng.pairing_batch(g1a, g2a)[0]
except Exception as e:   # noqa: BLE001
    res['pairing_legs_error'] = repr(e)
eng.timing_enable(True); eng.verify_batch(sig, [msg], [pk]); tm = eng.timing_read(); eng.timing_enable(False)
res['verify_kernels_ms'] = {k: round(v[0], 4) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:12]}
print('LATENCY_AB', tag, json.dumps(res), flush=True)
