"""Adversarial parity tests (-m gpu).

The engine keeps every stored value as a small multiple of p without conditional subtractions and accumulates up to eight limb products
per 64-bit column before one Montgomery reduction; the magnitude bounds that make this safe are tracked by the host compiler in floating
point (csrc/trace.cpp).  These tests feed the entry points that accept ARBITRARY field elements -- nbls_final_exp_batch, nbls_pairing_batch
and nbls_miller_product without validation -- with extremal operands (every coefficient p-1, 0 / 1 / p-1 mixtures, (p-1)/2, 2^380 + ...,
alternating limb patterns) and compare bit for bit with the oracle, through both Miller code paths (one program / LINES + ACC); then
16,384 pairwise distinct random pairs, and BASELINE configs[2] at full size: verifyBatch of 65,536 signatures, true and every way false.
Reference counterparts: test/fp12.test.ts (extremal field elements), test/index.test.ts:337-398 (verifyBatch true / wrong message / wrong key)."""
import hashlib
import importlib
import random
import pytest
from goldenio import hx

pytestmark = pytest.mark.gpu
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
NEVER_SPLIT = 1 << 40


@pytest.fixture(scope='module')
def eng():
    pkg = importlib.import_module('noble-bls12-381_amd')
    return pkg.Engine(0)


def fp(v):
    return (v % P).to_bytes(48, 'big')


# extremal field elements: the ends of the range, the middle, single high bits, alternating 28-bit limb patterns (the engine's limb size)
def _extremes():
    lim = (1 << 28) - 1
    alt_a = sum((lim if i % 2 == 0 else 0) << (28 * i) for i in range(14)) % P
    alt_b = sum((lim if i % 2 == 1 else 0) << (28 * i) for i in range(14)) % P
    all_ones_limbs = sum(lim << (28 * i) for i in range(13))           # 13 saturated limbs, below p
    return [P - 1, 0, 1, (P - 1) // 2, (P + 1) // 2, (1 << 380) + 0x123456789abcdef, (1 << 380) - 1, P - 2, 2, alt_a, alt_b, all_ones_limbs,
            (1 << 379) + (1 << 28) - 1, P - (1 << 28), 3 * (P // 4)]


EXT = _extremes()


def _fp12_cases():
    cases = []
    for v in EXT:
        if v:
            cases.append([v] * 12)                                      # every coefficient the same extreme
    cases.append([P - 1 if i % 2 == 0 else 0 for i in range(12)])       # alternating extremes
    cases.append([0 if i % 2 == 0 else P - 1 for i in range(12)])
    cases.append([P - 1, 1] * 6)
    cases.append([(P - 1) // 2, P - 1, 1, 0, P - 2, 2] * 2)
    cases.append([0] * 11 + [P - 1])                                    # a single non-zero coefficient, at either end
    cases.append([P - 1] + [0] * 11)
    rnd = random.Random(12)
    for _ in range(40):                                                 # random mixtures of the extremes
        cases.append([rnd.choice(EXT) for _ in range(12)])
    return [c for c in cases if any(c)]


def test_final_exp_extremes(eng, oracle):
    cases = _fp12_cases()
    blob = b''.join(b''.join(fp(v) for v in c) for c in cases)
    out = eng.final_exp_batch(blob)
    for i, c in enumerate(cases):
        e = blob[576 * i:576 * (i + 1)]
        assert out[576 * i:576 * (i + 1)] == oracle.un('fp12_final_exp', e, 576), (i, [hex(v)[:12] for v in c])


@pytest.mark.parametrize('n', [1, 7, 8, 9, 41, 300])
def test_final_exp_compressed_squarings(eng, oracle, n):
    """The compressed-squaring form of the five cyclotomic exponentiations (Karabina; csrc/pipelines_pairing.cpp expx, forced with NBLS_TUNE_EXPC_MIN = 0) on the
    extremal Fp12 inputs, random ones, and the inputs whose compressed coordinates vanish (the unit element, elements of Fp6 and of Fp2: their easy
    part is 1) scattered over the batch -- those are flagged on the device and recomputed by the plain program.  Batch sizes around the wavefront fills
    (8 items per wavefront in the squaring program, 5 in the decompression).  Every result against the oracle (math.ts:856-874)."""
    rnd = random.Random(1000 + n)
    cases = _fp12_cases()
    one = fp(1) + bytes(528)
    items = []
    for i in range(n):
        k = i % 11
        if k == 3: items.append(one)
        elif k == 6: c = cases[rnd.randrange(len(cases))]; c = c if any(c[:6]) else [2] * 6; items.append(b''.join(fp(v) for v in c[:6]) + bytes(288))      # a non-zero element of Fp6
        elif k == 9: items.append(fp(rnd.randrange(1, P)) + fp(rnd.randrange(P)) + bytes(480))                            # an element of Fp2
        elif k % 2: items.append(b''.join(fp(rnd.randrange(P)) for _ in range(12)))
        else: items.append(b''.join(fp(v) for v in cases[rnd.randrange(len(cases))]))
    blob = b''.join(items)
    eng.set_expc_min(0)
    try:
        out = eng.final_exp_batch(blob)
    finally:
        eng.set_expc_min(1 << 40)
    assert out == eng.final_exp_batch(blob)                      # the plain program (the default)
    for i in range(n):
        assert out[576 * i:576 * (i + 1)] == oracle.un('fp12_final_exp', items[i], 576), (n, i, i % 11)


def test_pairings_with_compressed_squarings(eng, oracle):
    """9000 pairings in one call with the compressed squarings switched on: the call runs as two halves on two streams (two redo lists and counters in
    flight at once); all results against the multi-threaded oracle"""
    n = 9000
    rnd = random.Random(9)
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    P64 = [oracle.g1_mul(g1, rnd.randrange(1, 2 ** 250))[1] for _ in range(48)]; Q64 = [oracle.g2_mul(g2, rnd.randrange(1, 2 ** 250))[1] for _ in range(48)]
    G1 = b''.join(P64[i % 48] for i in range(n)); G2 = b''.join(Q64[(i // 48 + 5 * i) % 48] for i in range(n))
    eng.set_expc_min(0)
    try:
        out, st = eng.pairing_batch(G1, G2, True, False)
    finally:
        eng.set_expc_min(1 << 40)
    ref, _ = oracle.pairing_batch(G1, G2, True, False, threads=64)
    assert out == ref


def _coordinate_cases():
    """(G1, G2) wire pairs whose coordinates are arbitrary field elements (not curve points): the Miller loop is polynomial arithmetic and
    must agree with the oracle on any input"""
    rnd = random.Random(34)
    pairs = []
    for v in EXT:
        pairs.append((fp(v) * 2, fp(v) * 4))
    pairs.append((fp(P - 1) + fp(1), fp(0) + fp(P - 1) + fp(P - 1) + fp(0)))
    pairs.append((fp(1) + fp(P - 1), fp(P - 1) + fp(0) + fp(0) + fp(P - 1)))
    for _ in range(49):
        pairs.append((b''.join(fp(rnd.choice(EXT)) for _ in range(2)), b''.join(fp(rnd.choice(EXT)) for _ in range(4))))
    return pairs


@pytest.mark.parametrize('split_min', [NEVER_SPLIT, 0], ids=['one-program', 'lines+acc'])
def test_miller_extreme_coordinates(eng, oracle, split_min):
    pairs = _coordinate_cases()
    G1 = b''.join(p[0] for p in pairs); G2 = b''.join(p[1] for p in pairs)
    eng.set_split_miller_min(split_min)
    try:
        ml, _ = eng.pairing_batch(G1, G2, False, False)
        ref = [oracle.miller_loop(a, b) for a, b in pairs]
        for i in range(len(pairs)):
            assert ml[576 * i:576 * (i + 1)] == ref[i], i
        # with the final exponentiation wherever the Miller value is invertible (a zero value has no inverse in the reference either)
        keep = [i for i in range(len(pairs)) if any(ref[i])]
        out, _ = eng.pairing_batch(b''.join(pairs[i][0] for i in keep), b''.join(pairs[i][1] for i in keep), True, False)
        for k, i in enumerate(keep):
            assert out[576 * k:576 * (k + 1)] == oracle.un('fp12_final_exp', ref[i], 576), i
        # the shared-accumulator product program on the same inputs (odd and even counts)
        for m in (len(pairs), len(pairs) - 1, 2, 1):
            assert eng.miller_product(G1[:96 * m], G2[:192 * m], False)[0] == oracle.miller_product(G1[:96 * m], G2[:192 * m], final_exp=False), m
    finally:
        eng.set_split_miller_min(16384)


def _distinct_points(eng, oracle, n, seed):
    rnd = random.Random(seed)
    ks = [rnd.randrange(1, R) for _ in range(n)]
    Pts, st = eng.point_mul_batch([k.to_bytes(32, 'big') for k in ks]); assert not any(st)
    Qts, st = eng.point_mul_batch([((k * 7 + 3) % R or 1).to_bytes(32, 'big') for k in ks], pts=oracle.g2_generator() * n, g2=True); assert not any(st)
    return ks, Pts, Qts


@pytest.mark.parametrize('split_min', [NEVER_SPLIT, 0], ids=['one-program', 'lines+acc'])
def test_random_distinct_pairs(eng, oracle, split_min):
    """2 x 4096 pairwise distinct uniformly random (P, Q) per code path, with and without the final exponentiation, against the oracle on all
    host threads; bilinearity of the Miller product over each batch"""
    eng.set_split_miller_min(split_min)
    try:
        for seed in (1000, 1001):
            n = 4096
            ks, Pts, Qts = _distinct_points(eng, oracle, n, seed + (7 if split_min == 0 else 0))
            for fe in (True, False):
                out, _ = eng.pairing_batch(Pts, Qts, fe, False)
                ref, _ = oracle.pairing_batch(Pts, Qts, fe, False, threads=64)
                assert out == ref, (seed, fe)
            t = sum(k * ((k * 7 + 3) % R or 1) for k in ks) % R
            lhs = eng.miller_product(Pts, Qts, True)[0]
            rhs = eng.pairing_batch(oracle.g1_mul(oracle.g1_generator(), t)[1], oracle.g2_generator(), True, False)[0]
            assert lhs == rhs, seed
    finally:
        eng.set_split_miller_min(16384)


def test_verify_batch_65536(eng, oracle, golden):
    """BASELINE configs[2] at full size: one aggregate signature over 65,536 distinct messages (test/index.test.ts:337-398 at scale):
    true; one flipped message; one wrong key; two keys swapped; the identity as a key; an undecodable key (the reference throws)"""
    n = 65536
    pkg = importlib.import_module('noble-bls12-381_amd')
    sks = [(int.from_bytes(hashlib.sha256(b'nbls-test-sk' + i.to_bytes(4, 'big')).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'test-msg' + i.to_bytes(4, 'big')).digest() for i in range(n)]
    pks, sig = oracle.aggregate_sign(msgs, sks, threads=128)
    assert eng.verify_batch(sig, msgs, pks) is True
    bad = list(msgs); bad[n - 3] = bytes([bad[n - 3][0] ^ 1]) + bad[n - 3][1:]
    assert eng.verify_batch(sig, bad, pks) is False
    wrong = list(pks); wrong[12345] = oracle.get_public_key((7).to_bytes(32, 'big'))
    assert eng.verify_batch(sig, msgs, wrong) is False
    swapped = list(pks); swapped[1], swapped[2] = swapped[2], swapped[1]
    assert eng.verify_batch(sig, msgs, swapped) is False
    ident = list(pks); ident[n // 2] = bytes([0xc0]) + bytes(47)                     # PointG1.ZERO: pairing() throws, verifyBatch answers false
    assert eng.verify_batch(sig, msgs, ident) is False
    undec = [hx(v['hex']) for v in golden['codec']['g1'] if v['result'] == 'Invalid compressed G1 point'][0]
    broken = list(pks); broken[n - 1] = undec
    with pytest.raises(pkg.NblsError, match='decode'):
        eng.verify_batch(sig, msgs, broken)
    off_group = [hx(v['hex']) for v in golden['codec']['g1'] if 'subgroup' in v['result']][0]     # decodes, fails assertValidity: the reference throws as well
    broken[n - 1] = off_group
    with pytest.raises(pkg.NblsError, match='decode'):
        eng.verify_batch(sig, msgs, broken)
    # a signature that is not the aggregate
    other = oracle.sign(msgs[0], sks[0])[1]
    assert eng.verify_batch(other, msgs, pks) is False
