"""Compiled step programs executed by the host VM simulator, checked bit-for-bit against oracle/ (CPU only)."""
import ctypes as C
import pytest
import vmsim_py
from goldenio import hx


@pytest.fixture(scope='module')
def sim():
    return vmsim_py.load()


def _points(golden, n):
    g1 = b''.join(hx(v['g1']) for v in golden['pairs'][:n])
    g2 = b''.join(hx(v['g2']) for v in golden['pairs'][:n])
    return g1, g2


def test_miller_bytes(sim, oracle, golden):
    n = 3
    g1, g2 = _points(golden, n)
    out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'MILLER_BYTES', n, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 2: (out, 576)})
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['miller']), i


def test_full_pairing_pipeline(sim, oracle, golden):
    n = 4   # fe_hard runs 3 instances per wave: 4 items exercise a partially filled second wave
    g1, g2 = _points(golden, n)
    F = C.create_string_buffer(vmsim_py.F12 * n)
    N = C.create_string_buffer(vmsim_py.RAW * n)
    out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'MILLER_FE', n, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 3: (F, vmsim_py.F12), 4: (N, vmsim_py.RAW)})
    vmsim_py.final_exp(sim, n, F, N, out)
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['pairing']), i


def test_final_exp_and_product(sim, oracle, golden, testdata):
    # final_exp_batch path: wire bytes -> F, N -> inverse -> hard part
    fin = hx(testdata['finalexp_in']) + hx(golden['fp12'][0]['a'])
    n = 2
    F = C.create_string_buffer(vmsim_py.F12 * n); N = C.create_string_buffer(vmsim_py.RAW * n)
    out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'NORM_BYTES', n, {2: (C.create_string_buffer(fin, len(fin)), 576), 3: (F, vmsim_py.F12), 4: (N, vmsim_py.RAW)})
    vmsim_py.final_exp(sim, n, F, N, out)
    assert out.raw[:576] == hx(testdata['finalexp_out'])
    assert out.raw[576:] == hx(golden['fp12'][0]['finalexp'])
    # product of two Miller values then shared final exponentiation (golden 'product')
    p = golden['product']
    g1 = hx(''.join(p['g1'])); g2 = hx(''.join(p['g2']))
    F2 = C.create_string_buffer(vmsim_py.F12 * 2); Fp = C.create_string_buffer(vmsim_py.F12); outb = C.create_string_buffer(576)
    vmsim_py.run(sim, 'MILLER_RAW', 2, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 3: (F2, vmsim_py.F12)})
    vmsim_py.run(sim, 'MUL2', 1, {3: (F2, 2 * vmsim_py.F12), 5: (Fp, vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', 1, {3: (Fp, vmsim_py.F12), 2: (outb, 576)})
    assert outb.raw == hx(p['miller_product'])
    # the in-place level of the product tree (round 5): A (buf 3) * B (buf 4) -> buf 5 aliasing buf 3
    F2b = C.create_string_buffer(F2.raw, vmsim_py.F12 * 2)
    second = (C.c_char * vmsim_py.F12).from_buffer(F2b, vmsim_py.F12)
    vmsim_py.run(sim, 'MUL2S', 1, {3: (F2b, 2 * vmsim_py.F12), 4: (second, 2 * vmsim_py.F12), 5: (F2b, 2 * vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', 1, {3: (F2b, 2 * vmsim_py.F12), 2: (outb, 576)})
    assert outb.raw == hx(p['miller_product'])
    N1 = C.create_string_buffer(vmsim_py.RAW)
    vmsim_py.run(sim, 'NORM_RAW', 1, {3: (Fp, vmsim_py.F12), 4: (N1, vmsim_py.RAW)})
    vmsim_py.final_exp(sim, 1, Fp, N1, outb)
    assert outb.raw == hx(p['result'])


STATUS = {'ok': 0, 'Invalid G1 point: not on curve Fp': 2, 'Invalid G2 point: not on curve Fp2': 2,
          'Invalid G1 point: must be of prime-order subgroup': 3, 'Invalid G2 point: must be of prime-order subgroup': 3}


def test_validity_programs(sim, golden):
    for grp, prog, buf, sz in (('g1', 'G1_VALIDATE', 0, 96), ('g2', 'G2_VALIDATE', 1, 192)):
        vs = golden['validity'][grp]
        pts = b''.join(hx(v['aff']) for v in vs)
        st = C.create_string_buffer(len(vs))
        vmsim_py.run(sim, prog, len(vs), {buf: (C.create_string_buffer(pts, len(pts)), sz), 7: (st, 1)})
        assert list(st.raw) == [STATUS[v['result']] for v in vs], grp


CODEC_STATUS = {'ok': 0, 'zero': 1, 'Invalid G1 point: must be of prime-order subgroup': 3, 'Invalid G2 point: must be of prime-order subgroup': 3,
                'Invalid compressed G1 point': 4, 'Failed to find a square root': 4}


def test_decompress_programs(sim, golden):
    vs = golden['codec']['g1']
    out, st = vmsim_py.g1_decompress(sim, b''.join(hx(v['hex']) for v in vs))
    for i, v in enumerate(vs):
        assert st[i] == CODEC_STATUS[v['result']], (i, v['result'])
        if st[i] == 0:
            assert out[96 * i:96 * (i + 1)] == hx(v['aff'])
        else:
            assert out[96 * i:96 * (i + 1)] == bytes(96)
    vs = golden['codec']['g2']
    out, st = vmsim_py.g2_decompress(sim, b''.join(hx(v['hex']) for v in vs))
    for i, v in enumerate(vs):
        assert st[i] == CODEC_STATUS[v['result']], (i, v['result'])
        if st[i] == 0:
            assert out[192 * i:192 * (i + 1)] == hx(v['aff'])


def test_hash_to_g2_program(sim, oracle, golden, testdata, ls2=False, norm=False):
    vs = golden['h2c']
    uni = b''.join(oracle.expand_message_xmd(hx(v['msg']), v['dst'].encode(), 256) for v in vs)
    out = vmsim_py.hash_to_g2(sim, uni, ls2, norm)
    for i, v in enumerate(vs):
        assert out[192 * i:192 * (i + 1)] == hx(v['aff']), i
    suite = testdata['h2c_g2_ro']          # RFC 9380 vectors held by test/hashToCurve.test.ts
    uni = b''.join(oracle.expand_message_xmd(hx(v['msg']), suite['dst'].encode(), 256) for v in suite['vectors'])
    out = vmsim_py.hash_to_g2(sim, uni, ls2, norm)
    for i, v in enumerate(suite['vectors']):
        e = hx(v['x1x0y1y0'])
        assert out[192 * i:192 * (i + 1)] == e[48:96] + e[0:48] + e[144:192] + e[96:144], i


def test_point_sum_programs(sim, oracle, golden):
    for g2, key, sz in ((False, 'g1pts', 96), (True, 'g2pts', 192)):
        pts = b''.join(hx(v['aff']) for v in golden[key])
        for n in (1, 2, 3, 5, 6):
            out, st = vmsim_py.point_sum(sim, pts[:sz * n], g2)
            ref_st, ref = (oracle.g2_sum if g2 else oracle.g1_sum)(pts[:sz * n])
            assert st == ref_st == 0 and out == ref, (g2, n)
        v = golden[key][0]
        assert vmsim_py.point_sum(sim, hx(v['aff']) + hx(v['affQ']), g2)[0] == hx(v['sum_aff'])


def test_fp_inverse_edge_cases(sim):
    """the per-lane inversion routine (fp_inv.h) on structured inputs: any representative below 2^392, powers of two,
    all-ones, values next to p, tiny and zero; expected a^-1 in the same Montgomery form (Fp.invert, math.ts:134-156)"""
    import random
    p = vmsim_py.P_MOD
    rnd = random.Random(381)
    xs = [rnd.randrange(1, p) for _ in range(500)] + [rnd.randrange(1, 1 << rnd.randrange(1, 392)) for _ in range(1500)]
    xs += [1 << i for i in range(392)] + [(1 << i) - 1 for i in range(1, 392)] + [p - (1 << i) for i in range(380)]
    xs += [1, 2, p - 1, 0, 5 * p + 3, (1 << 391) + 12345, p + 1, 2 * p - 1]
    inp = b''.join(b''.join(((x >> (28 * i)) & 0xfffffff).to_bytes(4, 'little') for i in range(14)) + bytes(8) for x in xs)
    src = C.create_string_buffer(inp, len(inp)); dst = C.create_string_buffer(len(inp))
    sim.nbls_sim_fp_inv(C.c_uint(len(xs)), src, dst)
    R = 1 << 392
    for k, x in enumerate(xs):
        o = dst.raw[64 * k:64 * k + 64]
        got = sum(int.from_bytes(o[4 * i:4 * i + 4], 'little') << (28 * i) for i in range(14))
        assert got < 2 * p
        assert got % p == ((pow(x % p, -1, p) * R * R) % p if x % p else 0), hex(x)
    # round 6: the same algorithm with one limb per lane (fp_inv_wide.h; device: nbls_fp_inv_wide_kernel, launches of a few thousand elements at most): the same inverses,
    # exact limbs, below 2.1 p, and no violated 32 / 64-bit assumption or non-uniform "row-uniform" value in the host model
    sim.nbls_sim_wide_violations.restype = C.c_ulong
    sim.nbls_sim_wide_violations()
    dst2 = C.create_string_buffer(len(inp))
    sim.nbls_sim_fp_inv_wide(C.c_uint(len(xs)), src, dst2)
    for k, x in enumerate(xs):
        o = dst2.raw[64 * k:64 * k + 64]
        limbs = [int.from_bytes(o[4 * i:4 * i + 4], 'little') for i in range(16)]
        assert all(l < (1 << 28) for l in limbs[:14]) and limbs[14] == 0 and limbs[15] == 0, hex(x)
        got = sum(l << (28 * i) for i, l in enumerate(limbs[:14]))
        assert got < 21 * p // 10
        assert got % p == ((pow(x % p, -1, p) * R * R) % p if x % p else 0), hex(x)
    assert sim.nbls_sim_wide_violations() == 0


def _raw(xs):
    return b''.join(b''.join(((x >> (28 * i)) & 0xfffffff).to_bytes(4, 'little') for i in range(14)) + bytes(8) for x in xs)


def _unraw(buf, k):
    o = buf[64 * k:64 * k + 64]
    return sum(int.from_bytes(o[4 * i:4 * i + 4], 'little') << (28 * i) for i in range(14))


def test_pow_chains(sim):
    """the fixed-exponent chains of the per-lane kernels (pow_exec.h: sliding windows over odd powers, a dedicated Fp squaring, signed Fp2 operands, the
    split a^e = (conj(a) a^11)^K a^tail of the two Fp2 exponents) executed on the host, against Python's pow() and against the plain square-and-multiply
    stand-in: Fp.sqrt's a^((p+1)/4) (math.ts:251-264), the SWU exponent (p-3)/4, Fp2.sqrt's (p^2+7)/16 and sqrt_div_fp2's (p^2-9)/16 (math.ts:521-538, 1196-1198)"""
    import random
    p = vmsim_py.P_MOD
    R = 1 << 392
    rnd = random.Random(5381)
    xs = [rnd.randrange(0, p) for _ in range(40)] + [0, 1, 2, p - 1, p - 2, (1 << 380), (1 << 381) - 1 - p, 3 * p + 5, 15 * p + 7, (p + 1) // 2]
    for which, e in ((0, (p + 1) // 4), (3, (p - 3) // 4)):
        src = C.create_string_buffer(_raw(xs), 64 * len(xs)); dst = C.create_string_buffer(64 * len(xs))
        sim.nbls_sim_fp_pow(C.c_uint(len(xs)), src, dst, which)
        for k, x in enumerate(xs):
            got = _unraw(dst.raw, k)
            a = (x * pow(R, -1, p)) % p                      # the element a raw value stands for
            assert got < 4 * p and got % p == (pow(a, e, p) * R) % p, (which, hex(x))
    # Fp2: elements (c0, c1), both components any representative below 16 p
    def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)
    def f2pow(a, e):
        r = (1, 0)
        for bit in bin(e)[2:]:
            r = f2mul(r, r)
            if bit == '1': r = f2mul(r, a)
        return r
    pairs = [(rnd.randrange(0, p), rnd.randrange(0, p)) for _ in range(12)] + [(0, 0), (1, 0), (0, 1), (p - 1, p - 1), (5, 0), (0, 7), (15 * p + 3, 14 * p + 9), (p - 1, 1)]
    flat = [c for pr in pairs for c in pr]
    for which, e in ((1, (p * p + 7) // 16), (2, (p * p - 9) // 16)):
        src = C.create_string_buffer(_raw(flat), 64 * len(flat)); dst = C.create_string_buffer(64 * len(flat)); ref = C.create_string_buffer(64 * len(flat))
        sim.nbls_sim_fp_pow(C.c_uint(len(pairs)), src, dst, which)
        sim.nbls_sim_fp_pow_naive(C.c_uint(len(pairs)), src, ref, which)
        cd = C.create_string_buffer(64 * len(flat)); cr = C.create_string_buffer(64 * len(flat))
        sim.nbls_sim_fp_canon(C.c_uint(len(flat)), dst, cd); sim.nbls_sim_fp_canon(C.c_uint(len(flat)), ref, cr)
        assert cd.raw == cr.raw, which
        ri = pow(R, -1, p)
        for k, (x0, x1) in enumerate(pairs):
            want = f2pow(((x0 * ri) % p, (x1 * ri) % p), e)
            got = (_unraw(dst.raw, 2 * k), _unraw(dst.raw, 2 * k + 1))
            assert got[0] < 4 * p and got[1] < 4 * p
            assert (got[0] % p, got[1] % p) == ((want[0] * R) % p, (want[1] * R) % p), (which, k)
    # round 6: the same four exponents through the one-limb-per-lane form (pow_wide.h; the host runs the device's sequences with the cross-lane moves as loops and counts every
    # violated assumption -- a 64-bit column overflow, a carry out of a low-word addition, a borrow in a biased subtraction)
    sim.nbls_sim_fp_pow_wide.restype = C.c_ulong
    for which, e in ((0, (p + 1) // 4), (3, (p - 3) // 4)):
        src = C.create_string_buffer(_raw(xs), 64 * len(xs)); dst = C.create_string_buffer(64 * len(xs))
        assert sim.nbls_sim_fp_pow_wide(C.c_uint(len(xs)), src, dst, which) == 0
        for k, x in enumerate(xs):
            got = _unraw(dst.raw, k)
            a = (x * pow(R, -1, p)) % p
            assert got < 2 * p and got % p == (pow(a, e, p) * R) % p, ('wide', which, hex(x))
            assert all(int.from_bytes(dst.raw[64 * k + 4 * i:64 * k + 4 * i + 4], 'little') < (1 << 28) for i in range(14)) and dst.raw[64 * k + 56:64 * k + 64] == bytes(8)     # stored limbs are exact
    for which, e in ((1, (p * p + 7) // 16), (2, (p * p - 9) // 16)):
        src = C.create_string_buffer(_raw(flat), 64 * len(flat)); dst = C.create_string_buffer(64 * len(flat))
        assert sim.nbls_sim_fp_pow_wide(C.c_uint(len(pairs)), src, dst, which) == 0
        ri = pow(R, -1, p)
        for k, (x0, x1) in enumerate(pairs):
            want = f2pow(((x0 * ri) % p, (x1 * ri) % p), e)
            got = (_unraw(dst.raw, 2 * k), _unraw(dst.raw, 2 * k + 1))
            assert got[0] < 2 * p and got[1] < 2 * p
            assert (got[0] % p, got[1] % p) == ((want[0] * R) % p, (want[1] * R) % p), ('wide', which, k)
    # the chains are shorter than 4-bit fixed windows: squarings and multiplications of each op list
    for which, nbits in ((0, 379), (1, 377), (2, 377), (3, 379)):
        st = (C.c_uint * 2)()
        sim.nbls_sim_pow_ops(which, st)
        assert st[0] <= nbits and st[1] <= 70, (which, st[0], st[1])


@pytest.mark.parametrize('w3', [False, True])
def test_scalar_mul_programs(sim, oracle, golden, w3):
    """the double-and-add-always ladders behind getPublicKey / sign (index.ts:738-752) against the oracle's scalar
    multiplication: random and structured scalars (1, 2, r-1, r+1, 2^256-1, a value with a long zero run); both forms: 2-bit windows (launches deeper than one
    wavefront per SIMD) and 3-bit windows (the others)"""
    import random
    r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    rnd = random.Random(12381)
    ks = [1, 2, r - 1, r + 1, (1 << 256) - 1, 1 << 200, rnd.randrange(1, r), rnd.randrange(1, r)]
    g1 = oracle.g1_generator()
    p1 = hx(golden['g1pts'][0]) if isinstance(golden['g1pts'][0], str) else g1
    pts1 = b''.join([g1, p1] * 4)
    out, st = vmsim_py.point_mul(sim, pts1, b''.join(k.to_bytes(32, 'big') for k in ks), w3=w3)
    for i, k in enumerate(ks):
        assert st[i] == 0
        assert out[96 * i:96 * i + 96] == oracle.g1_mul(pts1[96 * i:96 * i + 96], k % r)[1], i
    g2 = oracle.g2_generator()
    q = oracle.g2_mul(g2, 0xabcdef123456789)[1]
    pts2 = b''.join([g2, q] * 4)
    out, st = vmsim_py.point_mul(sim, pts2, b''.join(k.to_bytes(32, 'big') for k in ks), g2=True, w3=w3)
    for i, k in enumerate(ks):
        assert st[i] == 0
        assert out[192 * i:192 * i + 192] == oracle.g2_mul(pts2[192 * i:192 * i + 192], k % r)[1], i
    # k = r: the result is the zero point (status 1), which the host wrapper reports as an invalid key (status 5)
    out, st = vmsim_py.point_mul(sim, g1, r.to_bytes(32, 'big'), w3=w3)
    assert st[0] == 1


def test_fixed_base_public_keys(sim, oracle, testdata):
    """getPublicKey without doublings (round 5, csrc/curve.h pt_mul_fixed_g1: [k]G1.BASE as a sum of table entries [d 2^(3 w)]G, all entries of a window read and one
    picked by masked selects) against the oracle's scalar multiplication and the reference's own keys (the private keys of test/index.test.ts's sign vectors):
    structured scalars -- single digits in the lowest, a middle and the short top window, all-ones, r - 1, r + 1, 2^255, 2^256 - 1 -- and random ones"""
    import random
    r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    rnd = random.Random(4711)
    ks = [1, 2, 7, 8, 1 << 3, 5 << 129, 1 << 255, (1 << 255) | 3, (1 << 256) - 1, r - 1, r + 1, int('49' * 32, 16)] + [rnd.randrange(1, r) for _ in range(4)]
    ks += [int(p, 16) for p, _, _ in testdata['sign_vectors'][:4]]
    tab = vmsim_py.g1_fixed_table(sim, oracle)
    g1 = oracle.g1_generator()
    out, st = vmsim_py.point_mul_fixed(sim, tab, b''.join(k.to_bytes(32, 'big') for k in ks))
    for i, k in enumerate(ks):
        assert st[i] == 0 and out[96 * i:96 * i + 96] == oracle.g1_mul(g1, k % r)[1], (i, hex(k))
    for i, (p, _, _) in enumerate(testdata['sign_vectors'][:4]):
        pk = oracle.get_public_key(bytes.fromhex(p.rjust(64, '0')))      # 48 bytes: x with the flag bits in the top three
        assert out[96 * (16 + i) + 1:96 * (16 + i) + 48] == pk[1:] and (out[96 * (16 + i)] & 0x1f) == (pk[0] & 0x1f)
    out, st = vmsim_py.point_mul_fixed(sim, tab, r.to_bytes(32, 'big') + (0).to_bytes(32, 'big'))      # k = r and k = 0: the zero point (status 1; the host wrapper reports status 5)
    assert st[0] == 1 and st[1] == 1


def test_gls_ladder_for_subgroup_points(sim, oracle):
    """sign's ladder with the key split along psi (round 5, csrc/codec.h pt_mul_gls_g2: k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3, one accumulator over the four digits) against the
    oracle's scalar multiplication on points of G2: structured keys (1, |z|, |z|^2, |z|^3 and their neighbours -- a digit rolling over --, r - 1, r + 1, 2^256 - 1: a3 of 65 bits) and random ones"""
    import random
    r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    Z = 0xd201000000010000
    rnd = random.Random(777)
    ks = [1, 2, 3, Z - 1, Z, Z + 1, Z * Z - 1, Z * Z, Z ** 3, Z ** 3 - 1, r - 1, r + 1, (1 << 256) - 1, (1 << 255) + 12345] + [rnd.randrange(1, r) for _ in range(2)]
    g2 = oracle.g2_generator()
    q = oracle.g2_mul(g2, 0x1234567890abcdef1234567)[1]
    pts = b''.join([g2, q] * (len(ks) // 2))
    out, st = vmsim_py.point_mul_gls(sim, pts, b''.join(k.to_bytes(32, 'big') for k in ks))
    for i, k in enumerate(ks):
        assert st[i] == 0 and out[192 * i:192 * i + 192] == oracle.g2_mul(pts[192 * i:192 * i + 192], k % r)[1], (i, hex(k))
    out, st = vmsim_py.point_mul_gls(sim, g2, r.to_bytes(32, 'big'))       # k = r: the zero point
    assert st[0] == 1


def test_sign_aligned_ladder_for_subgroup_points(sim, oracle, prog='G2_MUL_SAC'):
    """sign's ladder for launches of at most one wavefront per SIMD (round 5, csrc/codec.h pt_mul_sac_g2: the digits recoded sign-aligned, one addition per bit from a table of eight)
    against the oracle: keys whose a0 is even / odd, digits rolling over, extreme digits (a3 of 65 bits), r - 1, r + 1, r (the zero point), random ones"""
    import random
    r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    Z = 0xd201000000010000
    rnd = random.Random(778)
    ks = [1, 2, 3, 4, Z - 1, Z, Z + 1, Z * Z - 1, Z * Z, Z ** 3, Z ** 3 - 1, r - 1, r + 1, (1 << 256) - 1, (1 << 256) - 2, (1 << 255) + 12345, (Z - 1) * (1 + Z + Z * Z) + (Z ** 3) * ((1 << 64) - 1)] + [rnd.randrange(1, r) for _ in range(3)]
    g2 = oracle.g2_generator()
    q = oracle.g2_mul(g2, 0x1234567890abcdef1234567)[1]
    pts = b''.join([g2, q] * (len(ks) // 2))
    out, st = vmsim_py.point_mul_sac(sim, pts, b''.join(k.to_bytes(32, 'big') for k in ks), prog)
    for i, k in enumerate(ks):
        assert st[i] == 0 and out[192 * i:192 * i + 192] == oracle.g2_mul(pts[192 * i:192 * i + 192], k % r)[1], (i, hex(k))
    out, st = vmsim_py.point_mul_sac(sim, g2 + g2, r.to_bytes(32, 'big') + (2 * r).to_bytes(32, 'big'), prog)       # k = r, 2r: the zero point
    assert st[0] == 1 and st[1] == 1


def test_msm_pipeline(sim, oracle):
    """the bucket-method multi-scalar multiplication (dev_msm in csrc/pipelines_codec.cpp: sort by window digit, segmented sums,
    bit-sliced bucket weighting, Horner over the windows) with its step programs on the simulator, against the oracle's
    sum of scalar multiples; scalars with repeated digits (long runs in one bucket), zero digits and zero scalars"""
    import random
    rnd = random.Random(381)
    g1 = oracle.g1_generator()
    pts = [oracle.g1_mul(g1, rnd.randrange(1, 1 << 64))[1] for _ in range(9)]
    ks = [rnd.randrange(1, 1 << 24) for _ in range(5)] + [0x5005, 0x5005, 0, 0xfff000]
    out, st = vmsim_py.msm(sim, b''.join(pts), b''.join(k.to_bytes(32, 'big') for k in ks), 24)
    ref = oracle.g1_sum(b''.join(oracle.g1_mul(p, k)[1] for p, k in zip(pts, ks) if k))
    assert st == 0 and out == ref[1]
    # full-width scalars: 22 windows, incl. 2^256 - 1 and a multiple of the group order
    r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    kw = [rnd.randrange(0, 1 << 256) for _ in range(7)] + [(1 << 256) - 1, r]
    out, st = vmsim_py.msm(sim, b''.join(pts), b''.join(k.to_bytes(32, 'big') for k in kw), 256)
    ref = oracle.g1_sum(b''.join(oracle.g1_mul(p, k % r)[1] for p, k in zip(pts, kw) if k % r))
    assert st == 0 and out == ref[1]
    # G2, one window; the same point twice with k and 4096 - k... (digits sum to a multiple handled by the doublings)
    g2 = oracle.g2_generator()
    q = [oracle.g2_mul(g2, 3)[1], oracle.g2_mul(g2, 5)[1], g2]
    ks2 = [7, 4095, 100]
    out, st = vmsim_py.msm(sim, b''.join(q), b''.join(k.to_bytes(32, 'big') for k in ks2), 12, g2=True)
    assert st == 0 and out == oracle.g2_mul(g2, 3 * 7 + 5 * 4095 + 100)[1]
    # full-width scalars on G2: four points (-1)^i psi^i(Q) with the base-|z| digits of k
    kq = [rnd.randrange(0, 1 << 256), (1 << 256) - 1, r - 1]
    out, st = vmsim_py.msm(sim, b''.join(q), b''.join(k.to_bytes(32, 'big') for k in kq), 256, g2=True)
    assert st == 0 and out == oracle.g2_mul(g2, (3 * kq[0] + 5 * kq[1] + kq[2]) % r)[1]
    # the sum is the zero point: k P + k (-P)
    out, st = vmsim_py.msm(sim, pts[0] + oracle.un('g1_neg_aff', pts[0], 96), (5).to_bytes(32, 'big') * 2, 12)
    assert st == 1


def test_miller_shared_accumulator(sim, oracle, golden):
    """MILLER_RAW2: two pairs per item with one shared accumulator == product of the two separate Miller loops
    (the reference multiplies separate millerLoop values, index.ts:756-767, 810-817; same field element)"""
    n = 5   # 5 items = 10 pairs
    g1, g2 = _points(golden, 2 * n)
    F = C.create_string_buffer(vmsim_py.F12 * n)
    out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'MILLER_RAW2', n, {0: (C.create_string_buffer(g1, len(g1)), 192), 1: (C.create_string_buffer(g2, len(g2)), 384), 3: (F, vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', n, {3: (F, vmsim_py.F12), 2: (out, 576)})
    for i in range(n):
        ref = oracle.miller_product(g1[192 * i:192 * i + 192], g2[384 * i:384 * i + 384], final_exp=False)
        assert out.raw[576 * i:576 * i + 576] == ref, i


def test_partially_filled_waves(sim, oracle, golden):
    """7 items: EXPX runs 5 items per wave (12 lanes each), the Miller programs 4 -- both leave a partially filled last wave"""
    n = 7
    g1, g2 = _points(golden, n)
    F = C.create_string_buffer(vmsim_py.F12 * n); N = C.create_string_buffer(vmsim_py.RAW * n); out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'MILLER_FE', n, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 3: (F, vmsim_py.F12), 4: (N, vmsim_py.RAW)})
    vmsim_py.final_exp(sim, n, F, N, out)
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['pairing']), i


def test_compress_programs(sim, golden):
    """PointG1.toHex(true) / PointG2.toSignature (index.ts:359-371, 586-602) against the reference-generated codec vectors"""
    for g2 in (False, True):
        vs = [v for v in golden['codec']['g2' if g2 else 'g1'] if v['result'] == 'ok']
        assert len(vs) >= 4
        out = vmsim_py.compress(sim, b''.join(hx(v['aff']) for v in vs), g2)
        e = 96 if g2 else 48
        for i, v in enumerate(vs):
            assert out[e * i:e * i + e] == hx(v['hex']), (g2, i)


def test_g1_hash_and_encode_programs(sim, oracle, golden, testdata):
    """PointG1.hashToCurve / encodeToCurve and PointG2.encodeToCurve step programs (index.ts:331-350, 491-497) on the
    reference's known-answer blocks (test/hashToCurve.test.ts) and the reference-run vectors"""
    def g2wire(e):
        b = hx(e); return b[48:96] + b[0:48] + b[144:192] + b[96:144]
    for k in testdata['h2c_kats']:
        if (k['group'], k['kind']) == ('g2', 'hash'):
            continue
        msgs = [hx(v['msg']) for v in k['vectors']]
        ln = {('g1', 'hash'): 128, ('g1', 'encode'): 64, ('g2', 'encode'): 128}[(k['group'], k['kind'])]
        uni = b''.join(oracle.expand_message_xmd(m, k['dst'].encode(), ln) for m in msgs)
        if k['group'] == 'g1':
            out = vmsim_py.hash_to_g1(sim, uni, 2 if k['kind'] == 'hash' else 1)
            assert [out[96 * i:96 * i + 96] for i in range(len(msgs))] == [hx(v['expected']) for v in k['vectors']], k['suite']
        else:
            out = vmsim_py.encode_to_g2(sim, uni)
            assert [out[192 * i:192 * i + 192] for i in range(len(msgs))] == [g2wire(v['expected']) for v in k['vectors']], k['suite']
    vs = golden['h2c_more']
    uni = b''.join(oracle.expand_message_xmd(hx(v['msg']), v['dst'].encode(), 128) for v in vs)
    out = vmsim_py.hash_to_g1(sim, uni, 2)
    assert [out[96 * i:96 * i + 96] for i in range(len(vs))] == [hx(v['g1_hash']) for v in vs]


LINE_BYTES = 68 * 6 * vmsim_py.RAW


def _lines(sim, g1, g2, n, folded=True):
    L = C.create_string_buffer(LINE_BYTES * n)
    bufs = {1: (C.create_string_buffer(g2, len(g2)), 192), 3: (L, LINE_BYTES)}
    if folded:
        bufs[0] = (C.create_string_buffer(g1, len(g1)), 96)
    vmsim_py.run(sim, 'LINES_PQ' if folded else 'LINES_Q', n, bufs)
    return L


def test_line_tables(sim, golden):
    """LINES_Q + LINES_BYTES == calcPairingPrecomputes (math.ts:1331-1371): the 68 line triples of the reference, bit for bit
    (7 points: LINES runs 6 items per wave)"""
    import hashlib
    n = 7
    g1, g2 = _points(golden, n)
    L = _lines(sim, g1, g2, n, folded=False)
    out = C.create_string_buffer(19584 * n)
    vmsim_py.run(sim, 'LINES_BYTES', 68 * n, {3: (L, 6 * vmsim_py.RAW), 2: (out, 288)})
    for i, v in enumerate(golden['pairs'][:n]):
        t = out.raw[19584 * i:19584 * (i + 1)]
        assert t[:288] == hx(v['ell_first']) and t[-288:] == hx(v['ell_last']), i
        assert hashlib.sha256(t).hexdigest() == v['ell_sha256'], i
    # wire form back to raw: the round trip reproduces the table
    L2 = C.create_string_buffer(LINE_BYTES * n)
    vmsim_py.run(sim, 'LINES_FROM_BYTES', 68 * n, {2: (out, 288), 3: (L2, 6 * vmsim_py.RAW)})
    out2 = C.create_string_buffer(19584 * n)
    vmsim_py.run(sim, 'LINES_BYTES', 68 * n, {3: (L2, 6 * vmsim_py.RAW), 2: (out2, 288)})
    assert out2.raw == out.raw


def test_split_miller(sim, oracle, golden):
    """LINES_PQ + ACC_* == millerLoop / pairing of the reference (11 pairs: ACC runs 5 items per wave)"""
    n = 11
    g1, g2 = _points(golden, n)
    L = _lines(sim, g1, g2, n)
    out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'ACC_BYTES', n, {3: (L, LINE_BYTES), 2: (out, 576)})
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['miller']), i
    F = C.create_string_buffer(vmsim_py.F12 * n); N = C.create_string_buffer(vmsim_py.RAW * n)
    vmsim_py.run(sim, 'ACC_FE', n, {3: (L, LINE_BYTES), 5: (F, vmsim_py.F12), 4: (N, vmsim_py.RAW)})
    vmsim_py.final_exp(sim, n, F, N, out)
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['pairing']), i
    # two tables per item, shared accumulator == product of the separate Miller loops (index.ts:756-767, 810-817)
    m = n // 2
    F2 = C.create_string_buffer(vmsim_py.F12 * m)
    vmsim_py.run(sim, 'ACC2_RAW', m, {3: (L, 2 * LINE_BYTES), 5: (F2, vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', m, {3: (F2, vmsim_py.F12), 2: (out, 576)})
    for i in range(m):
        assert out.raw[576 * i:576 * i + 576] == oracle.miller_product(g1[192 * i:192 * i + 192], g2[384 * i:384 * i + 384], final_exp=False), i
    # four tables per item; the last group is filled up with unit tables (every line = 1), as nbls_miller_product_dev does
    unit = b''.join(vmsim_py.raw_elem(1) + bytes(5 * vmsim_py.RAW) for _ in range(68))
    assert len(unit) == LINE_BYTES
    g = (n + 3) // 4
    L4 = C.create_string_buffer(L.raw[:LINE_BYTES * n] + unit * (4 * g - n), LINE_BYTES * 4 * g)
    F4 = C.create_string_buffer(vmsim_py.F12 * g)
    vmsim_py.run(sim, 'ACC4_RAW', g, {3: (L4, 4 * LINE_BYTES), 5: (F4, vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', g, {3: (F4, vmsim_py.F12), 2: (out, 576)})
    for i in range(g):
        k = min(4, n - 4 * i)
        assert out.raw[576 * i:576 * i + 576] == oracle.miller_product(g1[384 * i:384 * i + 96 * k], g2[768 * i:768 * i + 192 * k], final_exp=False), i
    # eight tables per item (round 4: Miller products of 131,072 pairs and more), the last group filled up the same way
    g8 = (n + 7) // 8
    L8 = C.create_string_buffer(L.raw[:LINE_BYTES * n] + unit * (8 * g8 - n), LINE_BYTES * 8 * g8)
    F8 = C.create_string_buffer(vmsim_py.F12 * g8)
    vmsim_py.run(sim, 'ACC8_RAW', g8, {3: (L8, 8 * LINE_BYTES), 5: (F8, vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', g8, {3: (F8, vmsim_py.F12), 2: (out, 576)})
    for i in range(g8):
        k = min(8, n - 8 * i)
        assert out.raw[576 * i:576 * i + 576] == oracle.miller_product(g1[768 * i:768 * i + 96 * k], g2[1536 * i:1536 * i + 192 * k], final_exp=False), i
    # prepared lines (not folded) + G1: PointG1.millerLoop(Q) with Q.pairingPrecomputes() (index.ts:452-454, 703-711); one table shared by
    # every item (stride 0) pairs every P with the same Q
    LQ = _lines(sim, g1, g2, n, folded=False)
    FQ = C.create_string_buffer(vmsim_py.F12 * n)
    vmsim_py.run(sim, 'ACC_Q', n, {0: (C.create_string_buffer(g1, len(g1)), 96), 3: (LQ, LINE_BYTES), 5: (FQ, vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', n, {3: (FQ, vmsim_py.F12), 2: (out, 576)})
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['miller']), i
    vmsim_py.run(sim, 'ACC_Q', n, {0: (C.create_string_buffer(g1, len(g1)), 96), 3: (LQ, 0), 5: (FQ, vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', n, {3: (FQ, vmsim_py.F12), 2: (out, 576)})
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == oracle.miller_loop(g1[96 * i:96 * i + 96], g2[:192]), i


def test_wire_decoders(sim, golden2):
    """the round-2 codec programs on the reference-generated vectors (tools/gen_golden2.mjs): G2 fromHex(96), fromSignature(192), uncompressed forms, byte-order swap"""
    def status_ok(result, st):
        want = {'ok': {0}, 'zero': {1}, 'Failed to find a square root': {4}, 'Invalid compressed G2 point': {4, 7}}.get(result)
        if want is None:
            want = {2} if 'not on curve' in result else {3} if 'subgroup' in result else {6} if 'encoding flag' in result else {8}
        return st in want
    for key, kind, ln in (('g2_fromhex96', 'hex', 96), ('g2_fromsig192', 'sig192', 192)):
        vs = golden2[key]
        out, st = vmsim_py.g2_decompress(sim, b''.join(hx(v['hex']) for v in vs), mode=kind)
        for i, v in enumerate(vs):
            assert status_ok(v['result'], st[i]), (key, i, v['result'], st[i])
            assert out[192 * i:192 * (i + 1)] == (hx(v['aff']) if v['result'] == 'ok' else bytes(192)), (key, i)
    for key, prog, a in (('g1_raw96', 'G1_FROM_RAW', 96), ('g2_raw192', 'G2_FROM_RAW', 192)):
        vs = golden2[key]
        n = len(vs)
        inb = vmsim_py.buf(b''.join(hx(v['hex']) for v in vs)); out = vmsim_py.buf(a * n); st = vmsim_py.buf(n)
        vmsim_py.run(sim, prog, n, {0: (inb, a), 6: (out, a), 7: (st, 1)})
        for i, v in enumerate(vs):
            assert status_ok(v['result'], st.raw[i]), (key, i, v['result'], st.raw[i])
            assert out.raw[a * i:a * (i + 1)] == (hx(v['aff']) if v['result'] == 'ok' else bytes(a)), (key, i)
    vs = [v for v in golden2['g2_raw192'] if v['result'] == 'ok' and v['compressed']]      # Q.toHex(false) of valid points (canonical coordinates)
    inb = vmsim_py.buf(b''.join(hx(v['aff']) for v in vs)); out = vmsim_py.buf(192 * len(vs))
    vmsim_py.run(sim, 'G2_SWAP', len(vs), {0: (inb, 192), 2: (out, 192)})
    assert out.raw == b''.join(hx(v['hex']) for v in vs)


def test_uncompressed_flag_bits(sim, golden3):
    """Round-2 advisor finding: PointG2.fromHex applies its flag rules to 192-byte input too (index.ts:534-537, 563) -- 0x20 / 0x60 / 0xe0 in the
    first byte are 'Invalid encoding flag', the compression bit is 'Invalid point G2, expected 96/192 bytes', and a non-canonical x.c1 + k p whose
    value reaches bit 381 is rejected rather than reduced -- while PointG1.fromHex(96 B) looks at the infinity bit only.  Vectors: the reference
    itself (tools/gen_golden3.mjs), every flag combination over valid points, the zero encoding and garbage."""
    def want(result):
        return {'ok': 0, 'zero': 1}.get(result, 2 if 'not on curve' in result else 3 if 'subgroup' in result else 6 if 'encoding flag' in result else 8)
    for key, prog, a in (('g1_raw96_flags', 'G1_FROM_RAW', 96), ('g2_raw192_flags', 'G2_FROM_RAW', 192)):
        vs = golden3[key]
        n = len(vs)
        assert {v['flag'] for v in vs} == {0, 0x20, 0x40, 0x60, 0x80, 0xa0, 0xc0, 0xe0}
        inb = vmsim_py.buf(b''.join(hx(v['hex']) for v in vs)); out = vmsim_py.buf(a * n); st = vmsim_py.buf(n)
        vmsim_py.run(sim, prog, n, {0: (inb, a), 6: (out, a), 7: (st, 1)})
        for i, v in enumerate(vs):
            assert st.raw[i] == want(v['result']), (key, i, hex(v['flag']), v['result'], st.raw[i])
            assert out.raw[a * i:a * (i + 1)] == (hx(v['aff']) if v['result'] == 'ok' else bytes(a)), (key, i)
    assert {v['result'] for v in golden3['g2_raw192_flags']} >= {'ok', 'zero', 'Invalid encoding flag: 32', 'Invalid encoding flag: 96', 'Invalid encoding flag: 224', 'Invalid point G2, expected 96/192 bytes'}


@pytest.mark.parametrize('suffix', ['_LS', '_LS2'])
def test_lane_split_programs(sim, oracle, golden, suffix):
    """The latency variants (every K_DOT lane-op spread over four adjacent lanes, one item per wavefront; round 5: over TWO lanes, two items per wavefront -- launches of
    1025 .. 2048 items): same results as the golden pairs -- Miller value, and the whole pairing with the lane-split Miller and exponentiation programs."""
    n = 3
    g1, g2 = _points(golden, n)
    out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'MILLER_BYTES' + suffix, n, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 2: (out, 576)})
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['miller']), i
    F = C.create_string_buffer(vmsim_py.F12 * n); N = C.create_string_buffer(vmsim_py.RAW * n)
    vmsim_py.run(sim, 'MILLER_FE' + suffix, n, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 3: (F, vmsim_py.F12), 4: (N, vmsim_py.RAW)})
    vmsim_py.final_exp(sim, n, F, N, out, expx='EXPX' + suffix)
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['pairing']), i
    Fr = C.create_string_buffer(vmsim_py.F12 * n); outr = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'MILLER_RAW' + suffix, n, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 3: (Fr, vmsim_py.F12)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', n, {3: (Fr, vmsim_py.F12), 2: (outr, 576)})
    for i in range(n):
        assert outr.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['miller']), i


def test_compressed_exponentiation(sim, golden, testdata):
    """Karabina's compressed squarings (round 3; EXPC_SQ -> EXPC_DEC_A -> inversion -> EXPC_DEC_B, csrc/pipelines_pairing.cpp expx): the same final exponentiation,
    bit for bit, on the reference's finalExponentiate known answer (test/pairing.test.ts:65-96), reference-run Fp12 values, and the inputs whose
    compressed coordinates vanish -- the unit element and an element of Fp6 (its easy part is 1): those are flagged and recomputed by the plain program"""
    one = (1).to_bytes(48, 'big') + bytes(528)
    sub = hx(golden['fp12'][1]['a'])[:288] + bytes(288)            # c1 = 0: an element of Fp6, f^(p^6) / f = 1
    fin = [hx(testdata['finalexp_in'])] + [hx(v['a']) for v in golden['fp12'][:5]] + [one, sub] + [hx(v['miller']) for v in golden['pairs'][:3]]
    n = len(fin)       # 11 items: EXPC_SQ runs 8 per wavefront, the decompression programs 5
    F = C.create_string_buffer(vmsim_py.F12 * n); N = C.create_string_buffer(vmsim_py.RAW * n)
    out = C.create_string_buffer(576 * n); ref = C.create_string_buffer(576 * n)
    blob = b''.join(fin)
    vmsim_py.run(sim, 'NORM_BYTES', n, {2: (C.create_string_buffer(blob, len(blob)), 576), 3: (F, vmsim_py.F12), 4: (N, vmsim_py.RAW)})
    vmsim_py.final_exp(sim, n, F, N, out, expx='EXPC')
    vmsim_py.final_exp(sim, n, F, N, ref)
    assert out.raw == ref.raw
    assert out.raw[:576] == hx(testdata['finalexp_out'])
    for i in range(5):
        assert out.raw[576 * (i + 1):576 * (i + 2)] == hx(golden['fp12'][i]['finalexp']), i
    assert out.raw[576 * 6:576 * 7] == one and out.raw[576 * 7:576 * 8] == one
    for i in range(3):
        assert out.raw[576 * (8 + i):576 * (9 + i)] == hx(golden['pairs'][i]['pairing']), i
    # the flags: exactly the two degenerate items, in every one of the five exponentiations' first stage
    T0 = C.create_string_buffer(vmsim_py.F12 * n); NI = C.create_string_buffer(vmsim_py.RAW * n); T1 = C.create_string_buffer(vmsim_py.F12 * n)
    sim.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    vmsim_py.run(sim, 'FE_EASY', n, {3: (F, vmsim_py.F12), 4: (NI, vmsim_py.RAW), 5: (T0, vmsim_py.F12)})
    assert vmsim_py.expx_compressed(sim, n, T0, T1) == [6, 7]


def test_plain_formula_switches():
    """NBLS_DBL_PLAIN / NBLS_MUL12_PLAIN / NBLS_CYCSQR_PLAIN select the formulas that the difference-of-squares doublings, the split Fp12 middle product and the tripled-state cyclotomic squaring replaced (DESIGN.md
    section 3.3; they exist for A/B timing).  Both must stay bit-exact: the tests that cover the point chains, the line tables and the Fp12 products run again
    in a fresh process with the switches set (the switches are read once per process)."""
    import os, subprocess, sys
    env = dict(os.environ, NBLS_DBL_PLAIN='1', NBLS_MUL12_PLAIN='1', NBLS_CYCSQR_PLAIN='1')
    sel = 'validity or hash_to_g2 or scalar_mul or full_pairing or split_miller or final_exp or compressed_exponentiation'
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider', os.path.abspath(__file__), '-k', sel], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'passed' in r.stdout


def test_scalar_splits_as_the_device_runs_them(sim):
    """csrc/scalar_split.h (compiled into msm_kernels.hip and into the simulator): the base-|z| digits of a scalar in both layouts (G2: four digits; G1: k mod z^2, k div z^2)
    and the sign-aligned recoding of sign's ladder against Python integers -- structured scalars (digits rolling over, extreme digits, 0, 2^256 - 1) and random ones"""
    import random
    Z = 0xd201000000010000
    r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    rnd = random.Random(4242)
    ks = [0, 1, 2, Z - 1, Z, Z + 1, Z * Z - 1, Z * Z, Z * Z + 1, Z ** 3 - 1, Z ** 3, r - 1, r, r + 1, (1 << 256) - 1, (1 << 256) - 2, 1 << 255, (Z - 1) * (1 + Z + Z * Z) + (Z ** 3) * ((1 << 64) - 1)]
    ks += [rnd.randrange(1 << 256) for _ in range(200)] + [rnd.randrange(r) for _ in range(200)]
    n = len(ks)
    blob = b''.join(k.to_bytes(32, 'big') for k in ks)
    for dims in (4, 2, 0):
        out = C.create_string_buffer(n * (128 if dims == 0 else 32 * dims))
        sim.nbls_sim_scalar_split(C.c_uint(n), C.c_uint(dims), blob, out)
        for i, k in enumerate(ks):
            if dims == 4:
                a, kk = [], k
                for _ in range(3):
                    a.append(kk % Z); kk //= Z
                a.append(kk)
                want = b''.join(x.to_bytes(32, 'big') for x in a)
                got = out.raw[128 * i:128 * i + 128]
            elif dims == 2:
                want = (k % (Z * Z)).to_bytes(32, 'big') + (k // (Z * Z)).to_bytes(32, 'big')
                got = out.raw[64 * i:64 * i + 64]
            else:
                want = b''.join(x.to_bytes(32, 'big') for x in vmsim_py.sac_recode(k))
                got = out.raw[128 * i:128 * i + 128]
            assert got == want, (dims, hex(k))


def test_hash_to_g2_norm_method(sim, oracle, golden, testdata):
    """hash-to-G2 with the SWU square root by the norm method (P_H2C_NA / NM / NB around two Fp exponentiations, codec.h swu_norm_*): the reference's points on its own
    vectors (test/hashToCurve.test.ts) and on the generated ones -- the root's sign is fixed by sgn0 (math.ts:1264), so any method that finds a root finds the same point"""
    test_hash_to_g2_program(sim, oracle, golden, testdata, False, True)
    vs = golden['h2c_more']
    uni = b''.join(oracle.expand_message_xmd(hx(v['msg']), v['dst'].encode(), 256) for v in vs)
    assert vmsim_py.hash_to_g2(sim, uni, False, True) == vmsim_py.hash_to_g2(sim, uni, False, False)


def test_norm_method_square_root_corner_cases(sim):
    """P_H2C_NM / P_H2C_NB fed directly: values a = u conj(v) with a vanishing imaginary part, where delta = (a0 + n) / 2 is zero for the root n = -a0 of the norm and the
    programs must fall back to n = a0; both characters of a0 / d (the root is real or purely imaginary); a = 0.  With num = den = 1 the point's Y coordinate is the
    root y itself: y^2 d == a, and sgn0(y) == sgn0(t) (math.ts:1264)."""
    import random
    p = vmsim_py.P_MOD; R = 1 << 392; RAW = vmsim_py.RAW
    rnd = random.Random(9380)
    mont = lambda v: (v % p) * R % p
    cases = []
    for k in range(12):
        a0 = rnd.randrange(1, p); d = rnd.randrange(1, p)
        n = (p - a0) if k % 2 == 0 else a0                       # both roots of the norm a0^2: the first makes delta vanish
        cases.append(((a0, 0), d, n))
    cases.append(((0, 0), rnd.randrange(1, p), 0))
    for k in range(6):                                          # ordinary values whose norm is a square, with its root given either sign
        y = (rnd.randrange(p), rnd.randrange(p)); d = rnd.randrange(1, p)
        a = ((y[0] * y[0] - y[1] * y[1]) * d % p, 2 * y[0] * y[1] * d % p)
        n = pow((a[0] * a[0] + a[1] * a[1]) % p, (p + 1) // 4, p)
        assert n * n % p == (a[0] * a[0] + a[1] * a[1]) % p
        cases.append((a, d, n if k % 2 else p - n))
    m = len(cases)
    ts = [(rnd.randrange(p), rnd.randrange(p)) for _ in range(m)]
    T = vmsim_py.buf(_raw([mont(x) for t in ts for x in t]))
    st = []
    for (a, d, n) in cases:
        st += [mont(3), mont(5), mont(1), 0, mont(1), 0, mont(a[0]), mont(a[1]), mont(d)] + [0] * 7      # zt2 (unused: the norm is a square), num = 1, den = 1, a, d
    St = vmsim_py.buf(_raw(st)); Nn = vmsim_py.buf(_raw([mont(n) for (_, _, n) in cases]))
    E, Pw, Pt2 = vmsim_py.buf(RAW * m), vmsim_py.buf(RAW * m), vmsim_py.buf(6 * RAW * m)
    St9 = (C.c_char * (16 * RAW * m - 9 * RAW)).from_buffer(St, 9 * RAW)
    vmsim_py.run(sim, 'H2C_NM', m, {3: (T, 2 * RAW), 4: (St, 16 * RAW), 5: (Nn, RAW), 6: (St9, 16 * RAW), 7: (E, RAW)})
    sim.nbls_sim_fp_pow(C.c_uint(m), E, Pw, 3)
    vmsim_py.run(sim, 'H2C_NB', m, {3: (T, 2 * RAW), 4: (St, 16 * RAW), 5: (Pw, RAW), 6: (Pt2, 6 * RAW)})
    Ri = pow(R, -1, p)
    sgn0 = lambda x: (x[0] % 2) or (x[0] == 0 and x[1] % 2)
    real = imag = 0
    for k, ((a, d, n), t) in enumerate(zip(cases, ts)):
        X = [_unraw(Pt2.raw, 6 * k + j) * Ri % p for j in range(6)]
        assert X[0] == 1 and X[1] == 0 and X[4] == 1 and X[5] == 0, k
        y = (X[2], X[3])
        assert ((y[0] * y[0] - y[1] * y[1]) * d % p, 2 * y[0] * y[1] * d % p) == a, k
        if a != (0, 0):
            assert bool(sgn0(y)) == bool(sgn0(t)), k
        if k < 12:
            real += y[1] == 0; imag += y[0] == 0
    assert real and imag and real + imag == 12
