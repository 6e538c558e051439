"""Thread safety of one engine context (include/nbls.h: "calls on one context are serialised internally"): host-level calls keep their
staging and scratch buffers for the whole call, so concurrent callers -- the N-API addon runs verifyBatch on libuv worker threads beside
synchronous calls on the same context -- must each get their own answer.  A valid and a forged verifyBatch, a pairing batch and a hash
batch race on ONE context from four threads (ctypes releases the GIL for the duration of a call)."""
import hashlib
import importlib
import threading
import pytest

pytestmark = pytest.mark.gpu


def test_racing_calls_on_one_context(oracle):
    pkg = importlib.import_module('noble-bls12-381_amd')
    eng = pkg.Engine(0)
    n = 257
    sks = [(int.from_bytes(hashlib.sha256(b'race-sk' + bytes([i & 255, i >> 8])).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'race-msg' + bytes([i & 255, i >> 8])).digest() for i in range(n)]
    pks, sig = oracle.aggregate_sign(msgs, sks, threads=16)
    forged = list(msgs); forged[5] = bytes(32)
    g1 = b''.join(oracle.g1_mul(oracle.g1_generator(), k)[1] for k in range(1, 34))
    g2 = b''.join(oracle.g2_mul(oracle.g2_generator(), k)[1] for k in range(1, 34))
    ref_pair, _ = oracle.pairing_batch(g1, g2, True, False)
    hm = [b'race-%d' % i for i in range(40)]
    ref_hash = eng.hash_to_g2_batch(hm)
    errors = []

    def run(name, fn, expect, reps):
        try:
            for k in range(reps):
                got = fn()
                if got != expect:
                    errors.append((name, k))
        except Exception as e:   # noqa: BLE001
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=run, args=('valid', lambda: eng.verify_batch(sig, msgs, pks), True, 12)),
          threading.Thread(target=run, args=('forged', lambda: eng.verify_batch(sig, forged, pks), False, 12)),
          threading.Thread(target=run, args=('pairing', lambda: eng.pairing_batch(g1, g2, True, False)[0], ref_pair, 12)),
          threading.Thread(target=run, args=('hash', lambda: eng.hash_to_g2_batch(hm), ref_hash, 12))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_racing_calls_on_one_multi_handle(oracle):
    """Round-2 finding: nbls_multi_* shared ONE partial buffer per context and ONE gather buffer per handle, so that two product calls
    racing on a handle (the N-API addon runs nbls_multi_verify_batch on libuv worker threads) could be finished with each other's
    partials -- a forged batch answered with a valid batch's verdict.  Every call now owns its partial / gather buffers
    (csrc/nbls_multi.cpp CallBuffers).  A valid and a forged verifyBatch, a Miller product with a known value and a pairing batch race
    on ONE handle opened on device 0 twice, twelve times each, every answer checked."""
    pkg = importlib.import_module('noble-bls12-381_amd')
    m = pkg.MultiEngine([0, 0])
    n = 129
    sks = [(int.from_bytes(hashlib.sha256(b'mrace-sk' + bytes([i & 255, i >> 8])).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'mrace-msg' + bytes([i & 255, i >> 8])).digest() for i in range(n)]
    pks, sig = oracle.aggregate_sign(msgs, sks, threads=16)
    forged = list(msgs); forged[n - 2] = bytes(32)
    g1 = b''.join(oracle.g1_mul(oracle.g1_generator(), k)[1] for k in range(1, 34))
    g2 = b''.join(oracle.g2_mul(oracle.g2_generator(), k)[1] for k in range(1, 34))
    ref_pair, _ = oracle.pairing_batch(g1, g2, True, False)
    ref_prod = oracle.miller_product(g1, g2, final_exp=True)
    errors = []

    def run(name, fn, expect, reps):
        try:
            for k in range(reps):
                got = fn()
                if got != expect:
                    errors.append((name, k))
        except Exception as e:   # noqa: BLE001
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=run, args=('valid', lambda: m.verify_batch(sig, msgs, pks), True, 12)),
          threading.Thread(target=run, args=('forged', lambda: m.verify_batch(sig, forged, pks), False, 12)),
          threading.Thread(target=run, args=('product', lambda: m.miller_product(g1, g2, True, False)[0], ref_prod, 12)),
          threading.Thread(target=run, args=('pairing', lambda: m.pairing_batch(g1, g2, True, False)[0], ref_pair, 12))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    m.close()
    assert not errors, errors
