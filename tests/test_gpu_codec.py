"""GPU parity tests for the rows of SURVEY 8(a) around the pairing: validity checks, decoders, hash-to-G2, sums, verifyBatch."""
import hashlib
import importlib
import pytest
from ctypes import c_int as C_int
from goldenio import hx

pytestmark = pytest.mark.gpu
STATUS = {'ok': 0, 'zero': 1, 'Invalid G1 point: not on curve Fp': 2, 'Invalid G2 point: not on curve Fp2': 2,
          'Invalid G1 point: must be of prime-order subgroup': 3, 'Invalid G2 point: must be of prime-order subgroup': 3,
          'Invalid compressed G1 point': 4, 'Failed to find a square root': 4}


@pytest.fixture(scope='module')
def eng():
    return importlib.import_module('noble-bls12-381_amd').Engine(0)


def test_validity(eng, golden):
    for g2 in (False, True):
        vs = golden['validity']['g2' if g2 else 'g1']
        assert eng.validate_batch(b''.join(hx(v['aff']) for v in vs), g2) == [STATUS[v['result']] for v in vs]


def test_pairing_with_validation(eng, oracle, golden):
    v1, v2 = golden['validity']['g1'], golden['validity']['g2']
    g1 = b''.join(hx(v['aff']) for v in v1)
    g2 = b''.join(hx(v['aff']) for v in v2)
    out, st = eng.pairing_batch(g1, g2, True, True)
    for i in range(len(v1)):
        rst, ref = oracle.pairing(g1[96 * i:96 * i + 96], g2[192 * i:192 * i + 192], True, True)
        assert st[i] == rst
        assert out[576 * i:576 * i + 576] == (ref if rst == 0 else bytes(576))


def test_validation_beside_the_loop_and_in_line(eng, oracle, golden):
    """nbls_pairing_batch / nbls_miller_product with validate = 1: up to 1024 pairs the validity programs run on the side streams beside the Miller loop, above that in
    line before it; the points go up once either way.  Same statuses, same zeroed slots, the product refused (NBLS_EDECODE) when any point is invalid."""
    pkg = importlib.import_module('noble-bls12-381_amd')
    v1, v2 = golden['validity']['g1'], golden['validity']['g2']
    bad1 = [hx(v['aff']) for v in v1 if STATUS[v['result']] != 0][:2]; bad2 = [hx(v['aff']) for v in v2 if STATUS[v['result']] != 0][:2]
    ok1 = [hx(v['aff']) for v in v1 if STATUS[v['result']] == 0 and hx(v['aff']) != bytes(96)][:3]
    ok2 = [hx(v['aff']) for v in v2 if STATUS[v['result']] == 0 and hx(v['aff']) != bytes(192)][:3]
    assert bad1 and bad2 and ok1 and ok2
    for n in (1, 5, 1024, 1025, 1500):
        P = [ok1[i % len(ok1)] for i in range(n)]; Q = [ok2[(i // 3) % len(ok2)] for i in range(n)]
        G1, G2 = b''.join(P), b''.join(Q)
        out, st = eng.pairing_batch(G1, G2, True, True)
        assert st == bytes(n) and out == eng.pairing_batch(G1, G2, True, False)[0]
        if n <= 5:
            assert out == oracle.pairing_batch(G1, G2, True, False)[0]
        prod = eng.miller_product(G1, G2, True, True)
        assert prod[1] == bytes(n) and prod[0] == eng.miller_product(G1, G2, True, False)[0]
        if n == 1: continue
        P[n // 2] = bad1[0]; Q[n - 1] = bad2[0]
        if n > 4: P[3] = bad1[-1]; Q[3] = bad2[-1]
        B1, B2 = b''.join(P), b''.join(Q)
        out2, st2 = eng.pairing_batch(B1, B2, True, True)
        for i in sorted({n // 2, n - 1, 3 if n > 4 else n - 1, 0}):
            rst, ref = oracle.pairing(B1[96 * i:96 * i + 96], B2[192 * i:192 * i + 192], True, True)
            assert st2[i] == rst, (n, i)
            assert out2[576 * i:576 * i + 576] == (ref if rst == 0 else bytes(576)), (n, i)
        assert sum(1 for c in st2 if c) == len({n // 2, n - 1} | ({3} if n > 4 else set()))
        assert bytes(x for i, x in enumerate(out2) if st2[i // 576] == 0) == bytes(x for i, x in enumerate(out) if st2[i // 576] == 0)
        with pytest.raises(pkg.NblsError):
            eng.miller_product(B1, B2, True, True)
        assert eng.miller_product(G1, G2, True, True)[0] == prod[0]      # the context is fine afterwards


def test_decompress(eng, oracle, golden, testdata):
    for g2 in (False, True):
        vs = golden['codec']['g2' if g2 else 'g1']
        out, st = eng.decompress_batch(b''.join(hx(v['hex']) for v in vs), g2)
        sz = 192 if g2 else 96
        for i, v in enumerate(vs):
            assert st[i] == STATUS[v['result']], (g2, i)
            assert out[sz * i:sz * (i + 1)] == (hx(v['aff']) if st[i] == 0 else bytes(sz))
    # zkcrypto compressed vectors (test/deterministic.test.ts:49-113), i*G for i = 1..255
    comp = b''.join(hx(testdata['zk_g1_compressed'][i]) for i in range(1, 256))
    out, st = eng.decompress_batch(comp, False)
    assert st == [0] * 255 and out == b''.join(hx(testdata['zk_g1_uncompressed'][i]) for i in range(1, 256))
    comp = b''.join(hx(testdata['zk_g2_compressed'][i]) for i in range(1, 200))
    out, st = eng.decompress_batch(comp, True)
    exp = b''
    for i in range(1, 200):
        z = hx(testdata['zk_g2_uncompressed'][i])
        exp += z[48:96] + z[0:48] + z[144:192] + z[96:144]
    assert st == [0] * 199 and out == exp
    assert eng.decompress_batch(hx(testdata['zk_g1_compressed'][0]), False)[1] == [1]


def test_hash_to_g2(eng, oracle, golden, testdata):
    d = {}
    for v in golden['h2c']:
        d.setdefault(v['dst'], []).append(v)
    for dst, vs in d.items():
        out = eng.hash_to_g2_batch([hx(v['msg']) for v in vs], dst.encode())
        for i, v in enumerate(vs):
            assert out[192 * i:192 * (i + 1)] == hx(v['aff'])
    suite = testdata['h2c_g2_ro']
    out = eng.hash_to_g2_batch([hx(v['msg']) for v in suite['vectors']], suite['dst'].encode())
    for i, v in enumerate(suite['vectors']):
        e = hx(v['x1x0y1y0'])
        assert out[192 * i:192 * (i + 1)] == e[48:96] + e[0:48] + e[144:192] + e[96:144]
    msgs = [hashlib.sha256(b'm%d' % i).digest()[: 1 + i % 32] for i in range(300)]
    out = eng.hash_to_g2_batch(msgs)
    for i in (0, 1, 63, 64, 65, 299):
        assert out[192 * i:192 * (i + 1)] == oracle.hash_to_g2(msgs[i])[1]


def test_point_sums(eng, oracle, golden):
    for g2, key, sz in ((False, 'g1pts', 96), (True, 'g2pts', 192)):
        pts = b''.join(hx(v['aff']) + hx(v['affQ']) for v in golden[key])
        for n in (1, 2, 3, 7, 12):
            out, st = eng.point_sum(pts[:sz * n], g2)
            rst, ref = (oracle.g2_sum if g2 else oracle.g1_sum)(pts[:sz * n])
            assert (st, out) == (rst, ref)


def test_verify_batch(eng, oracle, golden, testdata):
    vb = golden['verify_batch']
    msgs, pks = [hx(m) for m in vb['msgs']], [hx(p) for p in vb['pks']]
    assert eng.verify_batch(hx(vb['agg_sig']), msgs, pks) is True
    m2 = list(msgs); m2[2] = m2[2][:5] + bytes([m2[2][5] ^ 0x40]) + m2[2][6:]
    assert eng.verify_batch(hx(vb['agg_sig']), m2, pks) is False
    p2 = list(pks); p2[1] = pks[0]
    assert eng.verify_batch(hx(vb['agg_sig']), msgs, p2) is False
    # single signatures from the reference's sign KATs (test/index.test.ts:287-293): verify == verifyBatch with n = 1
    for priv, msg, sig in testdata['sign_vectors'][:6]:
        pk = oracle.get_public_key(hx(priv.rjust(64, '0')))
        assert eng.verify_batch(hx(sig), [hx(msg)], [pk]) is True
        assert eng.verify_batch(hx(sig), [hx(msg) + b'x'], [pk]) is False
    # larger batch built with the oracle: 64 signers, aggregate signature
    sks = [hashlib.sha256(b'sk%d' % i).digest() for i in range(64)]
    sks = [(int.from_bytes(s, 'big') % (2**254) + 1).to_bytes(32, 'big') for s in sks]
    msgs = [hashlib.sha256(b'msg%d' % i).digest() for i in range(64)]
    pks = [oracle.get_public_key(s) for s in sks]
    sigs = [oracle.sign(m, s)[1] for m, s in zip(msgs, sks)]
    agg = oracle.aggregate_signatures(sigs)[1]
    assert eng.verify_batch(agg, msgs, pks) is True
    assert eng.verify_batch(agg, msgs[::-1], pks) is False


def test_hash_to_g2_message_and_dst_lengths(eng, oracle):
    """device expand_message_xmd (xmd_kernel.hip): every message length across the SHA-256 block boundaries of b_0 and a
    few DST lengths incl. an oversize DST (> 255 bytes is replaced by its digest, RFC 9380 5.3.3) against the oracle"""
    msgs = [bytes((7 * i + j) & 0xff for j in range(i)) for i in range(0, 200)] + [b'\xff' * 1000, b'\x00' * 4096]
    # (round 6: for DSTs of up to 85 bytes the blocks of b_1 .. b_8 are laid out once -- one block up to 21 bytes, two from 22: both sides of both switches)
    for dst in (b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_', b'QUUX-V01-CS02-with-BLS12381G2_XMD:SHA-256_SSWU_RO_', b'd', b'a' * 21, b'b' * 22, b'c' * 85, b'e' * 86, b'x' * 255, b'y' * 300):
        sub = msgs if len(dst) == 43 else msgs[::17]
        out = eng.hash_to_g2_batch(sub, dst)
        for i, m in enumerate(sub):
            assert out[192 * i:192 * i + 192] == oracle.hash_to_g2(m, dst)[1], (len(m), len(dst))


def test_compress_on_device(eng, oracle, golden, testdata):
    """nbls_g1_compress_batch / nbls_g2_compress_batch: the reference-generated codec vectors, the zkcrypto compressed
    vectors (multiples of the generators), and decompress(compress(P)) == P on random points incl. both sign flags"""
    for g2 in (False, True):
        vs = [v for v in golden['codec']['g2' if g2 else 'g1'] if v['result'] == 'ok']
        out = eng.compress_batch(b''.join(hx(v['aff']) for v in vs), g2)
        e = 96 if g2 else 48
        assert [out[e * i:e * i + e] for i in range(len(vs))] == [hx(v['hex']) for v in vs]
    g1, g2g = oracle.g1_generator(), oracle.g2_generator()
    pts1 = b''.join(oracle.g1_mul(g1, k)[1] for k in range(1, 40))
    pts2 = b''.join(oracle.g2_mul(g2g, k)[1] for k in range(1, 40))
    c1, c2 = eng.compress_batch(pts1), eng.compress_batch(pts2, True)
    zk1 = hx(testdata['zk_g1_compressed']) if isinstance(testdata['zk_g1_compressed'], str) else b''.join(hx(x) for x in testdata['zk_g1_compressed'])
    zk2 = hx(testdata['zk_g2_compressed']) if isinstance(testdata['zk_g2_compressed'], str) else b''.join(hx(x) for x in testdata['zk_g2_compressed'])
    # the zkcrypto files start with the point at infinity, then 1*G, 2*G, ...
    assert c1 == zk1[48:48 * 40] and c2 == zk2[96:96 * 40]
    back1, st1 = eng.decompress_batch(c1)
    back2, st2 = eng.decompress_batch(c2, True)
    assert back1 == pts1 and back2 == pts2 and not any(st1) and not any(st2)
    assert any(c[0] & 0x20 for c in (c1[48 * i:48 * i + 48] for i in range(39))) and any(not (c[0] & 0x20) for c in (c1[48 * i:48 * i + 48] for i in range(39)))


def test_hash_and_encode_to_curve_kats(eng, oracle, golden, testdata):
    """PointG1.hashToCurve / encodeToCurve and PointG2.encodeToCurve through the C ABI (device SHA-256 with 64 / 128 / 256
    uniform bytes per message): every known-answer block of the reference's test/hashToCurve.test.ts, the reference-run
    vectors, and message lengths across the SHA-256 block boundaries against the oracle"""
    def g2wire(e):
        b = hx(e); return b[48:96] + b[0:48] + b[144:192] + b[96:144]
    for k in testdata['h2c_kats']:
        msgs = [hx(v['msg']) for v in k['vectors']]
        g2 = k['group'] == 'g2'
        out = eng.hash_to_curve_batch(msgs, k['dst'].encode(), g2=g2, encode=k['kind'] == 'encode')
        sz = 192 if g2 else 96
        assert [out[sz * i:sz * i + sz] for i in range(len(msgs))] == [g2wire(v['expected']) if g2 else hx(v['expected']) for v in k['vectors']], k['suite']
    by_dst = {}
    for v in golden['h2c_more']:
        by_dst.setdefault(v['dst'], []).append(v)
    for dst, vs in by_dst.items():
        msgs = [hx(v['msg']) for v in vs]
        assert eng.hash_to_curve_batch(msgs, dst.encode()) == b''.join(hx(v['g1_hash']) for v in vs)
        assert eng.hash_to_curve_batch(msgs, dst.encode(), encode=True) == b''.join(hx(v['g1_encode']) for v in vs)
        assert eng.hash_to_curve_batch(msgs, dst.encode(), g2=True, encode=True) == b''.join(hx(v['g2_encode']) for v in vs)
    msgs = [bytes((3 * i + j) & 0xff for j in range(i)) for i in range(0, 150, 7)] + [b'z' * 700]
    dst = b'QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_'
    out = eng.hash_to_curve_batch(msgs, dst)
    enc = eng.hash_to_curve_batch(msgs, dst, encode=True)
    enc2 = eng.hash_to_curve_batch(msgs, dst, g2=True, encode=True)
    for i, m in enumerate(msgs):
        assert out[96 * i:96 * i + 96] == oracle.hash_to_g1(m, dst)[1]
        assert enc[96 * i:96 * i + 96] == oracle.encode_to_g1(m, dst)[1]
        assert enc2[192 * i:192 * i + 192] == oracle.encode_to_g2(m, dst)[1]
    # outputs are in the prime-order subgroup
    assert eng.validate_batch(out, False) == [0] * len(msgs) and eng.validate_batch(enc2, True) == [0] * len(msgs)


def test_decode_validate_large_mixed_batch(eng, oracle):
    """3001 compressed G1 keys and 1001 compressed G2 signatures (ragged last wavefronts at 16 / 8 items per wave), valid and
    invalid encodings mixed: decode status and affine bytes against the oracle item by item; the decoded points through the
    validity programs again; and 1999 hash-to-G2 outputs against the oracle"""
    import hashlib
    import random
    rnd = random.Random(2024)
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    n1, n2 = 3001, 1001
    base1 = [oracle.call('g1_compress', 48, oracle.g1_mul(g1, rnd.randrange(1, 1 << 200))[1], C_int(0))[1] for _ in range(40)]
    base2 = [oracle.call('g2_compress', 96, oracle.g2_mul(g2, rnd.randrange(1, 1 << 200))[1], C_int(0))[1] for _ in range(24)]
    keys = []
    for i in range(n1):
        k = bytearray(base1[i % 40])
        if i % 7 == 3:
            k[47 - (i % 5)] ^= 1 + (i % 200)          # another x: no square root, or a point outside the subgroup, or (rarely) valid
        keys.append(bytes(k))
    sigs = []
    for i in range(n2):
        s = bytearray(base2[i % 24])
        if i % 5 == 2:
            s[95 - (i % 3)] ^= 1 + (i % 100)
        sigs.append(bytes(s))
    for comp, is_g2, sz in ((keys, False, 96), (sigs, True, 192)):
        out, st = eng.decompress_batch(b''.join(comp), g2=is_g2)
        ok_pts = []
        for i, c in enumerate(comp):
            rs, ro = oracle.call('g2_decompress' if is_g2 else 'g1_decompress', sz, c)
            assert st[i] == rs, (is_g2, i, st[i], rs)
            if rs == 0:
                assert out[sz * i:sz * i + sz] == ro, (is_g2, i)
                ok_pts.append(ro)
        assert len(ok_pts) > len(comp) // 2
        assert eng.validate_batch(b''.join(ok_pts), g2=is_g2) == [0] * len(ok_pts)
    msgs = [hashlib.sha256(b'h2c-large-%d' % i).digest()[:1 + i % 32] for i in range(1999)]
    out = eng.hash_to_g2_batch(msgs)
    for i in range(0, 1999, 37):
        assert out[192 * i:192 * i + 192] == oracle.hash_to_g2(msgs[i])[1], i
