"""Pins oracle/ (the CPU restatement) against (a) the reference's own known-answer vectors and
(b) vectors produced by running the reference itself (tests/golden, tools/gen_golden.py).  CPU only."""
import ctypes as C
import hashlib
import pytest
from goldenio import hx


def test_fp_tower(oracle, golden):
    o = oracle
    for v in golden['fp']:
        a, b = hx(v['a']), hx(v['b'])
        assert o.bin('fp_add', a, b, 48) == hx(v['add'])
        assert o.bin('fp_sub', a, b, 48) == hx(v['sub'])
        assert o.bin('fp_mul', a, b, 48) == hx(v['mul'])
        assert o.bin('fp_mul', a, a, 48) == hx(v['sqr'])
        if v['inv']:
            assert o.un('fp_inv', a, 48) == hx(v['inv'])
        ok, r = o.call('fp_sqrt', 48, hx(v['sqr']))
        assert ok == 1 and r == hx(v['sqrt_of_sqr'])
        ok, r = o.call('fp_sqrt', 48, a)
        assert (ok == 1) == (v['sqrt_of_a'] is not None)
        if ok:
            assert r == hx(v['sqrt_of_a'])
    for v in golden['fp2']:
        a, b = hx(v['a']), hx(v['b'])
        for op in ('add', 'sub', 'mul'):
            assert o.bin('fp2_' + op, a, b, 96) == hx(v[op]), op
        for op in ('sqr', 'inv', 'frob1', 'mulnr', 'mulB'):
            assert o.un('fp2_' + op, a, 96) == hx(v[op]), op
        ok, r = o.call('fp2_sqrt', 96, hx(v['sqr']))
        assert ok == 1 and r == hx(v['sqrt_of_sqr'])
        ok, r = o.call('fp2_sqrt', 96, a)
        assert (ok == 1) == (v['sqrt_of_a'] is not None)
        if ok:
            assert r == hx(v['sqrt_of_a'])
    for v in golden['fp6']:
        a, b = hx(v['a']), hx(v['b'])
        assert o.bin('fp6_mul', a, b, 288) == hx(v['mul'])
        assert o.un('fp6_sqr', a, 288) == hx(v['sqr'])
        assert o.un('fp6_inv', a, 288) == hx(v['inv'])
        assert o.un('fp6_mulnr', a, 288) == hx(v['mulnr'])
        assert o.call('fp6_mul_by_1', 288, a, hx(v['b1']))[1] == hx(v['mul1'])
        assert o.call('fp6_mul_by_01', 288, a, hx(v['b0']), hx(v['b1']))[1] == hx(v['mul01'])
        for k in range(1, 6):
            assert o.fp6_frob(a, k) == hx(v['frob'][k - 1]), k
    for v in golden['fp12']:
        a, b = hx(v['a']), hx(v['b'])
        assert o.bin('fp12_mul', a, b, 576) == hx(v['mul'])
        assert o.un('fp12_sqr', a, 576) == hx(v['sqr'])
        assert o.un('fp12_inv', a, 576) == hx(v['inv'])
        assert o.un('fp12_conj', a, 576) == hx(v['conj'])
        assert o.call('fp12_mul_by_014', 576, a, hx(v['o0']), hx(v['o1']), hx(v['o4']))[1] == hx(v['mul014'])
        for i, k in enumerate((1, 2, 3, 6)):
            assert o.fp12_frob(a, k) == hx(v['frob'][i]), k
        assert o.un('fp12_cyclotomic_sqr', hx(v['unitary']), 576) == hx(v['cyclosqr'])
        assert o.un('fp12_cyclotomic_exp_x', hx(v['unitary']), 576) == hx(v['cycloexp'])
        assert o.un('fp12_final_exp', a, 576) == hx(v['finalexp'])


def test_reference_kats_pairing_test_ts(oracle, testdata):
    """test/pairing.test.ts:46-96: e(G1,G2) from zkcrypto and the finalExponentiate KAT."""
    o = oracle
    st, e = o.pairing(o.g1_generator(), o.g2_generator())
    assert st == 0 and e == hx(testdata['e_G1_G2'])
    assert o.un('fp12_final_exp', hx(testdata['finalexp_in']), 576) == hx(testdata['finalexp_out'])


def test_pairing_vectors(oracle, golden):
    o = oracle
    for v in golden['pairs']:
        g1, g2 = hx(v['g1']), hx(v['g2'])
        out = C.create_string_buffer(19584)
        n = o.lib.oracle_calc_pairing_precomputes(g2, out)
        assert n == v['ell_len'] == 68
        assert hashlib.sha256(out.raw).hexdigest() == v['ell_sha256']
        assert out.raw[:288] == hx(v['ell_first']) and out.raw[-288:] == hx(v['ell_last'])
        assert o.miller_loop(g1, g2) == hx(v['miller'])
        st, e = o.pairing(g1, g2, True, True)
        assert st == 0 and e == hx(v['pairing'])
        st, e = o.pairing(g1, g2, False, False)
        assert st == 0 and e == hx(v['miller'])
    p = golden['product']
    assert o.miller_product(hx(''.join(p['g1'])), hx(''.join(p['g2'])), False) == hx(p['miller_product'])
    r = o.miller_product(hx(''.join(p['g1'])), hx(''.join(p['g2'])), True)
    assert r == hx(p['result']) == hx(p['e_pow_41'])
    st, e = o.pairing(o.g1_generator(), o.g2_generator())
    out = C.create_string_buffer(576)
    o.lib.oracle_fp12_pow_u64(e, C.c_uint64(41), out)
    assert out.raw == r


def test_kilic_1000_pairings(oracle, testdata):
    """test/deterministic.test.ts:34-46: e(i*G1, i*G2), i = 1..1000 (first 200 here; the whole list in the batch test)."""
    o = oracle
    g1, g2 = o.g1_generator(), o.g2_generator()
    G1s, G2s = [], []
    for i in range(1, 201):
        G1s.append(o.g1_mul(g1, i)[1])
        G2s.append(o.g2_mul(g2, i)[1])
    out, st = o.pairing_batch(b''.join(G1s), b''.join(G2s), True, True, threads=8)
    assert st == bytes(200)
    for i in range(200):
        assert out[576 * i:576 * (i + 1)] == hx(testdata['pairing_iG1_iG2'][i]), i


def test_points(oracle, golden):
    o = oracle
    for v in golden['g1pts']:
        P, Q = hx(''.join(v['P'])), hx(''.join(v['Q']))
        assert o.un('g1_double_proj', P, 144) == hx(''.join(v['dbl']))
        assert o.bin('g1_add_proj', P, Q, 144) == hx(''.join(v['add']))
        assert o.bin('g1_add_proj', P, P, 144) == hx(''.join(v['add_self']))
        st, r = o.g1_mul(hx(v['aff']), 0xd201000000010000)
        assert st == 0 and r == hx(v['mulx'])
        st, r = o.g1_sum(hx(v['aff']) + hx(v['affQ']))
        assert st == 0 and r == hx(v['sum_aff'])
    for v in golden['g2pts']:
        P, Q = hx(''.join(v['P'])), hx(''.join(v['Q']))
        assert o.un('g2_double_proj', P, 288) == hx(''.join(v['dbl']))
        assert o.bin('g2_add_proj', P, Q, 288) == hx(''.join(v['add']))
        assert o.bin('g2_add_proj', P, P, 288) == hx(''.join(v['add_self']))
        st, r = o.g2_mul(hx(v['aff']), 0xd201000000010000)
        assert st == 0 and r == hx(v['mulx'])
        assert o.un('g2_psi', hx(v['aff']), 192) == hx(v['psi'])
        assert o.un('g2_psi2', hx(v['aff']), 192) == hx(v['psi2'])
        st, r = o.g2_sum(hx(v['aff']) + hx(v['affQ']))
        assert st == 0 and r == hx(v['sum_aff'])
    cc = golden['clear_cofactor']
    assert o.un('g2_clear_cofactor', hx(cc['in']), 192) == hx(cc['out'])


STATUS = {'ok': 0, 'zero': 1, 'Invalid G1 point: not on curve Fp': 2, 'Invalid G2 point: not on curve Fp2': 2,
          'Invalid G1 point: must be of prime-order subgroup': 3, 'Invalid G2 point: must be of prime-order subgroup': 3,
          'Invalid compressed G1 point': 4, 'Failed to find a square root': 4}


def test_validity_and_codecs(oracle, golden):
    o = oracle
    for v in golden['validity']['g1']:
        assert o.lib.oracle_g1_validate(hx(v['aff'])) == STATUS[v['result']]
    for v in golden['validity']['g2']:
        assert o.lib.oracle_g2_validate(hx(v['aff'])) == STATUS[v['result']]
    for v in golden['codec']['g1']:
        st, aff = o.call('g1_decompress', 96, hx(v['hex']))
        assert st == STATUS[v['result']], v
        if st == 0:
            assert aff == hx(v['aff'])
            out = C.create_string_buffer(48)
            o.lib.oracle_g1_compress(aff, 0, out)
            assert out.raw == hx(v['hex'])
    for v in golden['codec']['g2']:
        st, aff = o.call('g2_decompress', 192, hx(v['hex']))
        assert st == STATUS[v['result']], v
        if st == 0:
            assert aff == hx(v['aff'])
            out = C.create_string_buffer(96)
            o.lib.oracle_g2_compress(aff, 0, out)
            assert out.raw == hx(v['hex'])


def test_zkcrypto_vectors(oracle, testdata):
    """test/deterministic.test.ts:49-113: i*G in compressed / uncompressed form, i = 0..999 (vector 0 is infinity)."""
    o = oracle
    g1, g2 = o.g1_generator(), o.g2_generator()
    for i in list(range(1, 40)) + [500, 999]:
        aff1 = o.g1_mul(g1, i)[1]
        assert aff1 == hx(testdata['zk_g1_uncompressed'][i])
        st, d = o.call('g1_decompress', 96, hx(testdata['zk_g1_compressed'][i]))
        assert st == 0 and d == aff1
        aff2 = o.g2_mul(g2, i)[1]
        # zkcrypto uncompressed G2 = x.c1 || x.c0 || y.c1 || y.c0 (index.ts:623-629)
        z = hx(testdata['zk_g2_uncompressed'][i])
        assert aff2 == z[48:96] + z[0:48] + z[144:192] + z[96:144]
        st, d = o.call('g2_decompress', 192, hx(testdata['zk_g2_compressed'][i]))
        assert st == 0 and d == aff2
    assert o.call('g1_decompress', 96, hx(testdata['zk_g1_compressed'][0]))[0] == 1
    assert o.call('g2_decompress', 192, hx(testdata['zk_g2_compressed'][0]))[0] == 1


def test_hash_to_curve(oracle, golden, testdata):
    o = oracle
    assert o.expand_message_xmd(b'abc', b'QUUX-V01-CS02-with-expander-SHA256-128', 32) == hx(golden['xmd_abc_32'])
    for v in golden['h2c']:
        msg, dst = hx(v['msg']), v['dst'].encode()
        st, u = o.hash_to_field(msg, dst)
        assert st == 0 and u == hx(v['u'])
        if 'swu0' in v:
            st, s0 = o.call('map_to_curve_g2', 192, u[:96])
            assert st == 0 and s0 == hx(v['swu0'])
            assert o.un('isogeny_map_g2', s0, 192) == hx(v['iso0'])
        st, q = o.hash_to_g2(msg, dst)
        assert st == 0 and q == hx(v['aff'])
    # RFC 9380 G2 random-oracle suite (test/hashToCurve.test.ts:574-642); expected = x.c1||x.c0||y.c1||y.c0
    suite = testdata['h2c_g2_ro']
    for v in suite['vectors']:
        st, q = o.hash_to_g2(hx(v['msg']), suite['dst'].encode())
        e = hx(v['x1x0y1y0'])
        assert st == 0 and q == e[48:96] + e[0:48] + e[144:192] + e[96:144]


def test_signatures(oracle, golden, testdata):
    o = oracle
    for v in golden['sigs']:
        sk, msg, pk, sig = hx(v['sk']), hx(v['msg']), hx(v['pk']), hx(v['sig'])
        assert o.get_public_key(sk) == pk
        st, s = o.sign(msg, sk)
        assert st == 0 and s == sig
        assert o.verify(sig, msg, pk) == 1
        bad = bytes([msg[0] ^ 1]) + msg[1:]
        assert o.verify(sig, bad, pk) == 0
    vb = golden['verify_batch']
    msgs, pks = [hx(m) for m in vb['msgs']], [hx(p) for p in vb['pks']]
    assert o.aggregate_public_keys(pks) == (0, hx(vb['agg_pk']))
    assert o.aggregate_signatures([hx(s['sig']) for s in golden['sigs']]) == (0, hx(vb['agg_sig']))
    assert o.verify_batch(hx(vb['agg_sig']), msgs, pks) == 1
    m2 = list(msgs)
    m2[2] = m2[2][:5] + bytes([m2[2][5] ^ 0x40]) + m2[2][6:]
    assert o.verify_batch(hx(vb['agg_sig']), m2, pks) == 0
    p2 = list(pks)
    p2[1] = pks[0]
    assert o.verify_batch(hx(vb['agg_sig']), msgs, p2) == 0
    assert o.verify(hx(vb['same_sig']), hx(vb['same_msg']), hx(vb['agg_pk'])) == 1
    # the reference's sign KATs (test/index.test.ts:287-293), first 24 of 559
    for priv, msg, sig in testdata['sign_vectors'][:24]:
        st, s = o.sign(hx(msg), hx(priv.rjust(64, '0')))
        assert st == 0 and s == hx(sig)


def _g2_tohex_to_wire(e):
    """PointG2.toHex() order (x.c1 || x.c0 || y.c1 || y.c0) -> affine wire order (x.c0 || x.c1 || y.c0 || y.c1)"""
    b = hx(e)
    return b[48:96] + b[0:48] + b[144:192] + b[96:144]


def test_hash_and_encode_to_curve_kats(oracle, golden, testdata):
    """every hashToCurve / encodeToCurve known-answer block of the reference's test/hashToCurve.test.ts (RFC 9380 G1/G2 RO and
    NU suites, kilic's TESTGEN suites) and the reference-run vectors for G1 hash / G1 encode / G2 encode"""
    fn = {('g1', 'hash'): oracle.hash_to_g1, ('g1', 'encode'): oracle.encode_to_g1, ('g2', 'hash'): oracle.hash_to_g2, ('g2', 'encode'): oracle.encode_to_g2}
    n = 0
    for k in testdata['h2c_kats']:
        for v in k['vectors']:
            st, out = fn[(k['group'], k['kind'])](hx(v['msg']), k['dst'].encode())
            assert st == 0
            assert out == (hx(v['expected']) if k['group'] == 'g1' else _g2_tohex_to_wire(v['expected'])), (k['suite'], v['msg'][:16])
            n += 1
    assert n == 36
    for v in golden['h2c_more']:
        m, dst = hx(v['msg']), v['dst'].encode()
        assert oracle.hash_to_g1(m, dst)[1] == hx(v['g1_hash'])
        assert oracle.encode_to_g1(m, dst)[1] == hx(v['g1_encode'])
        assert oracle.encode_to_g2(m, dst)[1] == hx(v['g2_encode'])
