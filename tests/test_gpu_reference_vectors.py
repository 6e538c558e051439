"""The reference's WHOLE known-answer sets on the GPU (round-2 verdict: the engine was checked on sub-samples -- 128 / 255 / 199 -- while only
the CPU oracle saw all of them): test/deterministic.test.ts:34-46 (1000 kilic pairings e(i G1, i G2)) and :49-113 (4 x 1000 zkcrypto points
i G, i = 0..999, compressed and uncompressed, both groups), through the C ABI.  As in the reference's test, every vector is decoded, re-encoded,
and compared with [i]G computed independently (here nbls_g*_mul_batch)."""
import importlib
import pytest
from goldenio import hx

pytestmark = pytest.mark.gpu
N = 1000


@pytest.fixture(scope='module')
def eng():
    pkg = importlib.import_module('noble-bls12-381_amd')
    return pkg.Engine(0)


@pytest.fixture(scope='module')
def multiples(eng, oracle):
    """[i]G1, [i]G2 for i = 1..999 as affine wire bytes, from the engine's scalar multiplication (spot-checked against the oracle)"""
    ks = [i.to_bytes(32, 'big') for i in range(1, N)]
    P, st = eng.point_mul_batch(ks); assert not any(st)
    Q, st = eng.point_mul_batch(ks, pts=oracle.g2_generator() * (N - 1), g2=True); assert not any(st)
    for i in (1, 2, 500, 999):
        assert P[96 * (i - 1):96 * i] == oracle.g1_mul(oracle.g1_generator(), i)[1]
        assert Q[192 * (i - 1):192 * i] == oracle.g2_mul(oracle.g2_generator(), i)[1]
    return P, Q


def test_all_1000_kilic_pairings(eng, oracle, testdata, multiples):
    """test/deterministic.test.ts:34-46, every vector; vector i is e((i+1) G1, (i+1) G2)"""
    vs = testdata['pairing_iG1_iG2']
    assert len(vs) == N
    P, Q = multiples
    k1000 = (N).to_bytes(32, 'big')
    P += eng.point_mul_batch([k1000])[0]
    Q += eng.point_mul_batch([k1000], pts=oracle.g2_generator(), g2=True)[0]
    out, st = eng.pairing_batch(P, Q, True, True)
    assert not any(st)
    for i in range(N):
        assert out[576 * i:576 * (i + 1)] == hx(vs[i]), i


def _swap_g2(z):     # zkcrypto / toHex(false) order x.c1 x.c0 y.c1 y.c0 -> affine wire order
    return z[48:96] + z[0:48] + z[144:192] + z[96:144]


def test_all_zkcrypto_g1(eng, testdata, multiples):
    P, _ = multiples
    comp = [hx(v) for v in testdata['zk_g1_compressed']]
    unc = [hx(v) for v in testdata['zk_g1_uncompressed']]
    assert len(comp) == N and len(unc) == N
    # PointG1.fromHex on both forms: index 0 is the zero point, i >= 1 is [i]G1
    out_c, st_c = eng.decode_points('g1', b''.join(comp), 48)
    out_u, st_u = eng.decode_points('g1', b''.join(unc), 96)
    assert list(st_c) == [1] + [0] * (N - 1) and list(st_u) == [1] + [0] * (N - 1)
    assert out_c[96:] == P and out_u[96:] == P
    assert out_c[:96] == bytes(96) and out_u[:96] == bytes(96)
    # toHex(true) / toHex(false) of [i]G1 reproduce the vectors, zero included
    zero = [1] + [0] * (N - 1)
    assert eng.encode_points(bytes(96) + P, g2=False, compressed=True, zero=zero) == b''.join(comp)
    assert eng.encode_points(bytes(96) + P, g2=False, compressed=False, zero=zero) == b''.join(unc)
    # the decompression entry point of verify (48 B keys)
    out, st = eng.decompress_batch(b''.join(comp[1:]), False)
    assert st == [0] * (N - 1) and out == P


def test_all_zkcrypto_g2(eng, testdata, multiples):
    _, Q = multiples
    comp = [hx(v) for v in testdata['zk_g2_compressed']]
    unc = [hx(v) for v in testdata['zk_g2_uncompressed']]
    assert len(comp) == N and len(unc) == N
    for kind in ('g2', 'sig'):                      # PointG2.fromHex and PointG2.fromSignature on the 96-byte form
        out_c, st_c = eng.decode_points(kind, b''.join(comp), 96)
        assert list(st_c) == [1] + [0] * (N - 1), kind
        assert out_c[192:] == Q, kind
    # the uncompressed 192-byte form is PointG2.fromHex's alone (fromSignature reads 192 bytes as a compressed pair of 96-byte halves, index.ts:500-504)
    out_u, st_u = eng.decode_points('g2', b''.join(unc), 192)
    assert list(st_u) == [1] + [0] * (N - 1)
    assert out_u[192:] == Q
    zero = [1] + [0] * (N - 1)
    assert eng.encode_points(bytes(192) + Q, g2=True, compressed=True, zero=zero) == b''.join(comp)
    assert eng.encode_points(bytes(192) + Q, g2=True, compressed=False, zero=zero) == b''.join(unc)
    assert b''.join(_swap_g2(u) for u in unc[1:]) == Q
    out, st = eng.decompress_batch(b''.join(comp[1:]), True)
    assert st == [0] * (N - 1) and out == Q


def test_doubling_kats_and_wnaf_scalars(eng, oracle, testdata):
    """test/point.test.ts:89-146, 270-346: the projective doubling known answers (compared as group elements: the engine returns affine points);
    :347-372 the wNAF scalar list: [k]G against [k-1]G + G and against the oracle"""
    P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    inv = lambda v: pow(v, P_MOD - 2, P_MOD)
    for v in testdata['point_double_kats']['g1']:
        x, y, z = (int(c, 16) for c in v['a']); X, Y, Z = (int(c, 16) for c in v['double'])
        aff = lambda a, b, c: (a * inv(c) % P_MOD).to_bytes(48, 'big') + (b * inv(c) % P_MOD).to_bytes(48, 'big')
        a = aff(x, y, z)
        dbl, st = eng.point_sum(a + a)
        assert st == 0 and dbl == aff(X, Y, Z)
        assert eng.point_mul_batch([(2).to_bytes(32, 'big')], pts=a)[0] == dbl
        assert eng.validate_batch(dbl) == [0]
    def f2inv(c0, c1):
        n = inv((c0 * c0 + c1 * c1) % P_MOD)
        return c0 * n % P_MOD, (-c1 * n) % P_MOD
    def f2mul(a, b):
        return (a[0] * b[0] - a[1] * b[1]) % P_MOD, (a[0] * b[1] + a[1] * b[0]) % P_MOD
    for v in testdata['point_double_kats']['g2']:
        c = [int(t, 16) for t in v['a']]; d = [int(t, 16) for t in v['double']]
        def aff2(t):
            zi = f2inv(t[4], t[5]); x = f2mul((t[0], t[1]), zi); y = f2mul((t[2], t[3]), zi)
            return b''.join(w.to_bytes(48, 'big') for w in (x[0], x[1], y[0], y[1]))
        a = aff2(c)
        dbl, st = eng.point_sum(a + a, g2=True)
        assert st == 0 and dbl == aff2(d)
        assert eng.point_mul_batch([(2).to_bytes(32, 'big')], pts=a, g2=True)[0] == dbl
        assert eng.validate_batch(dbl, g2=True) == [0]
    ks = [int(k, 16) for k in testdata['wnaf_scalars']]
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    A, st = eng.point_mul_batch([k.to_bytes(32, 'big') for k in ks]); assert not any(st)
    B, st = eng.point_mul_batch([(k - 1).to_bytes(32, 'big') for k in ks]); assert not any(st)
    A2, st = eng.point_mul_batch([k.to_bytes(32, 'big') for k in ks], pts=g2 * len(ks), g2=True); assert not any(st)
    for i, k in enumerate(ks):
        assert A[96 * i:96 * i + 96] == oracle.g1_mul(g1, k)[1]
        assert eng.point_sum(B[96 * i:96 * i + 96] + g1)[0] == A[96 * i:96 * i + 96]
        assert A2[192 * i:192 * i + 192] == oracle.g2_mul(g2, k)[1]
