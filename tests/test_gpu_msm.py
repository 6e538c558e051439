"""GPU parity tests of the multi-scalar multiplication (nbls_g1_msm / nbls_g2_msm) against the CPU oracle."""
import hashlib
import importlib
import random

import pytest

pytestmark = pytest.mark.gpu
R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


@pytest.fixture(scope='module')
def eng():
    pkg = importlib.import_module('noble-bls12-381_amd')
    return pkg.Engine(0)


def _points(oracle, n, seed, g2=False):
    """n points a_i * G with known a_i (so that large sums can be checked with ONE oracle multiplication)"""
    gen = oracle.g2_generator() if g2 else oracle.g1_generator()
    mul = oracle.g2_mul if g2 else oracle.g1_mul
    a = [int.from_bytes(hashlib.sha256(b'msm-%d-%d' % (seed, i)).digest(), 'big') % R_ORDER or 1 for i in range(n)]
    return a, [mul(gen, x)[1] for x in a]


def _ref(oracle, a, ks, g2=False):
    gen = oracle.g2_generator() if g2 else oracle.g1_generator()
    t = sum(x * k for x, k in zip(a, ks)) % R_ORDER
    return (oracle.g2_mul if g2 else oracle.g1_mul)(gen, t)[1] if t else None


@pytest.mark.parametrize('n', [1, 2, 3, 33, 500])
def test_g1_msm_vs_oracle(eng, oracle, n):
    rnd = random.Random(n)
    a, pts = _points(oracle, n, n)
    ks = [rnd.randrange(0, 1 << 256) for _ in range(n)]
    out, st = eng.msm(b''.join(pts), [k.to_bytes(32, 'big') for k in ks])
    # the oracle's own sum of scalar multiples (independent of the a_i bookkeeping)
    ref = oracle.g1_sum(b''.join(oracle.g1_mul(p, k % R_ORDER)[1] for p, k in zip(pts, ks) if k % R_ORDER))
    assert st == 0 and out == ref[1]
    assert out == _ref(oracle, a, ks)


@pytest.mark.parametrize('n', [1, 5, 120])
def test_g2_msm_vs_oracle(eng, oracle, n):
    rnd = random.Random(100 + n)
    a, pts = _points(oracle, n, 100 + n, g2=True)
    ks = [rnd.randrange(0, 1 << 256) for _ in range(n)]
    out, st = eng.msm(b''.join(pts), [k.to_bytes(32, 'big') for k in ks], g2=True)
    ref = oracle.g2_sum(b''.join(oracle.g2_mul(p, k % R_ORDER)[1] for p, k in zip(pts, ks) if k % R_ORDER))
    assert st == 0 and out == ref[1]


def test_msm_edge_cases(eng, oracle):
    a, pts = _points(oracle, 40, 7)
    P = b''.join(pts)
    # empty sum, all-zero scalars, k P + k (-P): the zero point (status 1)
    assert eng.msm(b'', [])[1] == 1
    assert eng.msm(P, [bytes(32)] * 40)[1] == 1
    neg = oracle.un('g1_neg_aff', pts[0], 96)
    assert eng.msm(pts[0] + neg, [(12345).to_bytes(32, 'big')] * 2)[1] == 1
    # every point in the same bucket of every window (one run of length n: the segmented sum needs ceil(log2 n) rounds)
    k = 0x0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef
    out, st = eng.msm(P, [k.to_bytes(32, 'big')] * 40)
    assert st == 0 and out == _ref(oracle, a, [k] * 40)
    # short scalars (64 bit: 6 windows), scalars >= r, digits 0 and 4095
    rnd = random.Random(5)
    ks = [rnd.randrange(0, 1 << 64) for _ in range(40)]
    out, st = eng.msm(P, [x.to_bytes(32, 'big') for x in ks])
    assert st == 0 and out == _ref(oracle, a, ks)
    ks = [R_ORDER, R_ORDER + 1, (1 << 256) - 1, 0xfff, 0xfff000, 1] + [rnd.randrange(0, 1 << 256) for _ in range(34)]
    out, st = eng.msm(P, [x.to_bytes(32, 'big') for x in ks])
    assert st == 0 and out == _ref(oracle, a, ks)


def test_msm_large_linear(eng, oracle):
    """65,536 points (64 distinct base points a_j G repeated with different scalars): the result must be
    (sum_i a_i k_i mod r) G -- one oracle multiplication; and linearity: msm(P, k) + msm(P, k') == msm(P, k + k')"""
    n = 65536
    a64, p64 = _points(oracle, 64, 99)
    rnd = random.Random(65536)
    a = [a64[i % 64] for i in range(n)]
    P = b''.join(p64) * (n // 64)
    k1 = [rnd.randrange(0, R_ORDER) for _ in range(n)]
    k2 = [rnd.randrange(0, R_ORDER) for _ in range(n)]
    o1, s1 = eng.msm(P, [k.to_bytes(32, 'big') for k in k1])
    o2, s2 = eng.msm(P, [k.to_bytes(32, 'big') for k in k2])
    o3, s3 = eng.msm(P, [((x + y) % R_ORDER).to_bytes(32, 'big') for x, y in zip(k1, k2)])
    assert s1 == 0 and s2 == 0 and s3 == 0
    assert o1 == _ref(oracle, a, k1) and o2 == _ref(oracle, a, k2)
    assert oracle.g1_sum(o1 + o2)[1] == o3
    # G2 at 4096
    n2 = 4096
    b64, q64 = _points(oracle, 64, 98, g2=True)
    kk = [rnd.randrange(0, R_ORDER) for _ in range(n2)]
    o, s = eng.msm(b''.join(q64) * (n2 // 64), [k.to_bytes(32, 'big') for k in kk], g2=True)
    assert s == 0 and o == _ref(oracle, [b64[i % 64] for i in range(n2)], kk, g2=True)
