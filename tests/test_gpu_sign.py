"""GPU parity tests for the secret-scalar side: getPublicKey / sign (reference index.ts:738-752) through the
C ABI (nbls_g1_mul_batch, nbls_g2_mul_batch, nbls_sign_batch) against the reference's own sign vectors
(test/bls12-381/fixtures/bls12-381-g2-test-vectors.txt, repacked in tests/golden/ref_testdata.json.gz), reference-generated
key/signature triples and the oracle."""
import importlib
import random
import pytest
from goldenio import hx

pytestmark = pytest.mark.gpu
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


@pytest.fixture(scope='module')
def eng():
    return importlib.import_module('noble-bls12-381_amd').Engine(0)


def test_sign_vectors(eng, testdata):
    vs = testdata['sign_vectors']
    assert len(vs) == 559
    sigs = eng.sign_batch([hx(v[1]) for v in vs], [hx(v[0]) for v in vs])
    for v, s in zip(vs, sigs):
        assert s == hx(v[2]), v[0]


def test_public_keys_and_signatures_of_reference_run(eng, golden):
    vs = golden['sigs']
    assert eng.get_public_keys([hx(v['sk']) for v in vs]) == [hx(v['pk']) for v in vs]
    assert eng.sign_batch([hx(v['msg']) for v in vs], [hx(v['sk']) for v in vs]) == [hx(v['sig']) for v in vs]


def test_point_mul_structured_scalars(eng, oracle):
    rnd = random.Random(381)
    ks = [1, 2, 3, R - 1, R + 1, 2 * R + 5, (1 << 256) - 1, 1 << 255, 1 << 200, (1 << 128) - 1] + [rnd.randrange(1, R) for _ in range(54)]
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    p1 = [oracle.g1_mul(g1, rnd.randrange(1, R))[1] for _ in range(4)]
    p2 = [oracle.g2_mul(g2, rnd.randrange(1, R))[1] for _ in range(4)]
    sc = [k.to_bytes(32, 'big') for k in ks]
    out, st = eng.point_mul_batch(sc)                      # generator
    assert st == bytes(len(ks))
    for i, k in enumerate(ks):
        assert out[96 * i:96 * i + 96] == oracle.g1_mul(g1, k % R)[1]
    pts = b''.join(p1[i % 4] for i in range(len(ks)))
    out, st = eng.point_mul_batch(sc, pts)
    for i, k in enumerate(ks):
        assert out[96 * i:96 * i + 96] == oracle.g1_mul(p1[i % 4], k % R)[1]
    pts = b''.join(p2[i % 4] for i in range(len(ks)))
    out, st = eng.point_mul_batch(sc, pts, g2=True)
    assert st == bytes(len(ks))
    for i, k in enumerate(ks):
        assert out[192 * i:192 * i + 192] == oracle.g2_mul(p2[i % 4], k % R)[1]


def test_invalid_keys(eng):
    """normalizePrivKey (index.ts:269-279): keys that are 0 mod r are rejected"""
    pkg = importlib.import_module('noble-bls12-381_amd')
    for bad in (0, R, 2 * R):
        out, st = eng.point_mul_batch([bad.to_bytes(32, 'big'), (5).to_bytes(32, 'big')])
        assert st == bytes([5, 0])
        with pytest.raises(pkg.NblsError):
            eng.get_public_keys([bad.to_bytes(32, 'big')])
        with pytest.raises(pkg.NblsError):
            eng.sign_batch([b'msg'], [bad.to_bytes(32, 'big')])


def test_sign_then_verify_batch(eng):
    """signatures produced on the GPU verify on the GPU: aggregate of distinct-message signatures through verifyBatch"""
    rnd = random.Random(7)
    n = 33
    sks = [rnd.randrange(1, R).to_bytes(32, 'big') for _ in range(n)]
    msgs = [bytes([i]) * (i + 1) for i in range(n)]
    pks = eng.get_public_keys(sks)
    sigs = eng.sign_batch(msgs, sks)
    affs, st = eng.decompress_batch(b''.join(sigs), True)
    assert st == [0] * n
    agg, z = eng.point_sum(affs, g2=True)
    assert z == 0
    assert eng.verify_batch(eng.compress_g2(agg), msgs, pks) is True
    assert eng.verify_batch(eng.compress_g2(agg), msgs[::-1], pks) is False
