#!/usr/bin/env python3
"""Derive every BLS12-381 constant the engine and the oracle need from first principles
(p, r, the BLS parameter x, the generators, and the RFC 9380 isogeny/SWU parameters) and emit
them as C tables in Montgomery form (R = 2^384):

    oracle/consts_gen.h                              6 x u64 limbs  (CPU oracle, test infrastructure)
    noble-bls12-381_amd/csrc/consts_gen.h            12 x u32 limbs (HIP engine)

Reference for what each constant means: /root/reference/math.ts:10-53 (CURVE), 1411-1543
(Frobenius / roots-of-unity / eta tables), 1546-1610 (3-isogeny, = RFC 9380 appendix E.3).
The Frobenius tables, roots of unity, psi constants and Montgomery constants are COMPUTED here
(and cross-checked by tests/ against reference-generated vectors); only curve parameters and the
published RFC 9380 isogeny coefficients / etas are literals.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
X = 0xd201000000010000  # |x|; the BLS parameter is -X
G1X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
G2X = (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
       0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e)
G2Y = (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
       0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)

# p = (x-1)^2 (x^4 - x^2 + 1)/3 + x with x = -X ;  r = x^4 - x^2 + 1
assert R_ORDER == X ** 4 - X ** 2 + 1
assert P == ((-X - 1) ** 2 * R_ORDER) // 3 + (-X)

RBITS = 384
RMONT = 1 << RBITS


# ---- tiny Fp2 arithmetic on Python ints (tuples) ---------------------------------------------
def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2mul(r, a)
        a = f2mul(a, a)
        e >>= 1
    return r


def f2inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def f2conj(a):
    return (a[0], (-a[1]) % P)


XI = (1, 1)  # the sextic non-residue u + 1

# Frobenius coefficients (math.ts:1428-1543): gamma_k = xi^((p^k - 1)/6), k = 0..11
FROB12 = [f2pow(XI, (P ** k - 1) // 6) for k in range(12)]
FROB6_1 = [f2pow(XI, (P ** k - 1) // 3) for k in range(6)]
FROB6_2 = [f2pow(XI, (2 * P ** k - 2) // 3) for k in range(6)]
for k in range(6):
    assert FROB6_1[k] == f2mul(FROB12[k], FROB12[k])
    assert FROB6_2[k] == f2mul(FROB6_1[k], FROB6_1[k])
# 8th roots of unity (math.ts:1435): (1+u)^(k (p^2-1)/8)
ROOTS8 = [f2pow(XI, (P * P - 1) * k // 8) for k in range(8)]
assert ROOTS8[2] == (0, 1) and ROOTS8[4] == (P - 1, 0)
# etas for the 9-mod-16 SWU square-root (math.ts:1417-1424, 1447-1452; RFC 9380 / draft-11 G.2.3)
ev1 = 0x699be3b8c6870965e5bf892ad5d2cc7b0e85a117402dfd83b7f4a947e02d978498255a2aaec0ac627b5afbdf1bf1c90
ev2 = 0x8157cd83046453f5dd0972b6e3949e4288020b5b8a9cc99ca07e27089a2ce2436d965026adad3ef7baba37f2183e9b5
ev3 = 0xab1c2ffdd6c253ca155231eb3e71ba044fd562f6f72bc5bad5ec46a0b7a3b0247cf08ce6c6317f40edbc653a72dee17
ev4 = 0xaa404866706722864480885d68ad0ccac1967c7544b447873cc37e0181271e006df72162a3d3e0287bf597fbf7f8fc1
ETAS = [(ev1, ev2), ((-ev2) % P, ev1), (ev3, ev4), ((-ev4) % P, ev3)]
SWU_Z = ((-2) % P, (-1) % P)
SWU_A = (0, 240)
SWU_B = (1012, 1012)
# property check: eta^2 = Z^3 * zeta for a primitive 8th root of unity zeta
Z3 = f2mul(f2mul(SWU_Z, SWU_Z), SWU_Z)
for e in ETAS:
    zeta = f2mul(f2mul(e, e), f2inv(Z3))
    assert f2pow(zeta, 8) == (1, 0) and f2pow(zeta, 4) != (1, 0), 'eta check'

# psi (untwist-Frobenius-twist, math.ts:1390-1403) reduces to (conj(x) * PSI_X, conj(y) * PSI_Y)
PSI_X = f2inv(f2pow(XI, (P - 1) // 3))
PSI_Y = f2inv(f2pow(XI, (P - 1) // 2))
# psi^2 (math.ts:1406-1412): x * PSI2_C1, -y
PSI2_C1 = f2mul(PSI_X, f2conj(PSI_X))
assert PSI2_C1[1] == 0
assert PSI2_C1[0] == 0x1a0111ea397fe699ec02408663d4de85aa0d857d89759ad4897d29650fb85f9b409427eb4f49fffd8bfd00000000aaac
# G1 endomorphism phi (index.ts:425-429): cube root of unity
BETA = FROB6_1[2][0]
assert FROB6_1[2][1] == 0 and pow(BETA, 3, P) == 1 and BETA != 1
assert BETA == 0x5f19672fdf76ce51ba69c6076a0f77eaddb3a93be6f89688de17d813620a00022e01fffffffefffe

# 3-isogeny E2' -> E2, RFC 9380 appendix E.3 (math.ts:1547-1610); coefficient lists are highest degree first
K = 0x5c759507e8e333ebb5b7a9a47d7ed8532c52d39fd3a042a88b58423c50ae15d5c2638e343d9c71c6238aaaaaaaa97d6
ISO_XNUM = [
    (0x171d6541fa38ccfaed6dea691f5fb614cb14b4e7f4e810aa22d6108f142b85757098e38d0f671c7188e2aaaaaaaa5ed1, 0),
    (0x11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71e,
     0x8ab05f8bdd54cde190937e76bc3e447cc27c3d6fbd7063fcd104635a790520c0a395554e5c6aaaa9354ffffffffe38d),
    (0, 0x11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71a),
    (K, K),
]
ISO_XDEN = [(0, 0), (1, 0), (0xc, P - 0xc), (0, P - 0x48)]
ISO_YNUM = [
    (0x124c9ad43b6cf79bfbf7043de3811ad0761b0f37a1e26286b0e977c69aa274524e79097a56dc4bd9e1b371c71c718b10, 0),
    (0x11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71c,
     0x8ab05f8bdd54cde190937e76bc3e447cc27c3d6fbd7063fcd104635a790520c0a395554e5c6aaaa9354ffffffffe38f),
    (0, 0x5c759507e8e333ebb5b7a9a47d7ed8532c52d39fd3a042a88b58423c50ae15d5c2638e343d9c71c6238aaaaaaaa97be),
    (0x1530477c7ab4113b59a4c18b076d11930f7da5d4a07f649bf54439d87d27e500fc8c25ebf8c92f6812cfc71c71c6d706,
     0x1530477c7ab4113b59a4c18b076d11930f7da5d4a07f649bf54439d87d27e500fc8c25ebf8c92f6812cfc71c71c6d706),
]
ISO_YDEN = [(1, 0), (0x12, P - 0x12), (0, P - 0xd8), (P - 0x1b0, P - 0x1b0)]


def _iso_check():
    """Evaluate the isogeny on a point of E2': y^2 = x^3 + 240u x + 1012(1+u) and check it lands on E2."""
    def add(a, b):
        return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)

    def horner(cs, x):
        acc = cs[0]
        for c in cs[1:]:
            acc = add(f2mul(acc, x), c)
        return acc
    t = 3
    while True:
        x = (t, 1)
        rhs = add(add(f2mul(f2mul(x, x), x), f2mul(SWU_A, x)), SWU_B)
        # sqrt in Fp2 via exponent (p^2+7)/16 and 8th roots
        cand = f2pow(rhs, (P * P + 7) // 16)
        y = None
        for rt in ROOTS8:
            c = f2mul(cand, rt)
            if f2mul(c, c) == rhs:
                y = c
                break
        if y is not None:
            break
        t += 1
    xn, xd, yn, yd = (horner(c, x) for c in (ISO_XNUM, ISO_XDEN, ISO_YNUM, ISO_YDEN))
    X2 = f2mul(xn, f2inv(xd))
    Y2 = f2mul(y, f2mul(yn, f2inv(yd)))
    lhs = f2mul(Y2, Y2)
    rhs2 = add(f2mul(f2mul(X2, X2), X2), (4, 4))
    assert lhs == rhs2, 'isogeny constants do not map E2\' to E2'


_iso_check()

# exponents
EXP = {
    'P_MINUS_2': P - 2,
    'P_PLUS_1_DIV_4': (P + 1) // 4,
    'P2_PLUS_7_DIV_16': (P * P + 7) // 16,      # Fp2.sqrt candidate exponent (math.ts:493, ORDER = p^2 - 1)
    'P2_MINUS_9_DIV_16': (P * P - 9) // 16,     # sqrt_div_fp2 (math.ts:1191)
}

N0_64 = (-pow(P, -1, 1 << 64)) % (1 << 64)
N0_32 = (-pow(P, -1, 1 << 32)) % (1 << 32)
assert N0_64 == 0x89f3fffcfffcfffd and N0_32 == 0xfffcfffd


def mont(v, rbits=RBITS):
    return (v % P) * (1 << rbits) % P


def limbs(v, bits, rbits=RBITS):
    n = rbits // bits
    return [(v >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]


def c_fp(v, bits, raw=False, rbits=RBITS):
    w = limbs(v if raw else mont(v, rbits), bits, rbits)
    fmt = '0x%016xull' if bits == 64 else '0x%08xu'
    return '{' + ','.join(fmt % x for x in w) + '}'


def emit(path, bits, fp_t, guard, qual, rbits=RBITS):
    L = []
    a = L.append
    a('/* GENERATED by tools/gen_consts.py -- do not edit.  Montgomery form, R = 2^%d, %d-bit limbs, little-endian limb order. */' % (rbits, bits))
    a('#ifndef %s\n#define %s' % (guard, guard))
    word = 'uint64_t' if bits == 64 else 'uint32_t'
    n = rbits // bits
    RM = 1 << rbits
    if bits == 28:
        # bias for limb-wise subtraction: 16p with every limb >= 2^28 - 1 (limb i borrows 2^28 from limb i+1), see vm_exec.h
        c = limbs(16 * P, 28, rbits)
        b16 = [c[0] + (1 << 28)] + [c[i] + (1 << 28) - 1 for i in range(1, n - 1)] + [c[n - 1] - 1]
        assert sum(x << (28 * i) for i, x in enumerate(b16)) == 16 * P and all(x >= (1 << 28) - 1 for x in b16[:-1])
        a('%s uint32_t NBLS_BIAS16[%d] = {%s};' % (qual, n, ','.join('0x%08xu' % x for x in b16)))
        a('#define NBLS_BIAS16_INIT {%s}' % ','.join('0x%08xu' % x for x in b16))
        a('#define NBLS_P_INIT {%s}' % ','.join('0x%08xu' % x for x in limbs(P, 28, rbits)))
        a('#define NBLS_R3_INIT {%s}' % ','.join('0x%08xu' % x for x in limbs(pow(RM, 3, P), 28, rbits)))
        a('#define NBLS_P_WORDS_INIT {%s}' % ','.join('0x%08xu' % ((P >> (32 * i)) & 0xffffffff) for i in range(12)))

    def fp(name, v, raw=False):
        a('%s %s %s[%d] = %s;' % (qual, word, name, n, c_fp(v, bits, raw, rbits)))

    def fp2(name, v):
        a('%s %s %s[2][%d] = {%s,%s};' % (qual, word, name, n, c_fp(v[0], bits, False, rbits), c_fp(v[1], bits, False, rbits)))

    def fp2arr(name, vs):
        a('%s %s %s[%d][2][%d] = {' % (qual, word, name, len(vs), n))
        for v in vs:
            a('  {%s,%s},' % (c_fp(v[0], bits, False, rbits), c_fp(v[1], bits, False, rbits)))
        a('};')

    def exp(name, e):
        nw = (e.bit_length() + 63) // 64
        a('#define NBLS_%s_BITS %d' % (name, e.bit_length()))
        a('%s uint64_t NBLS_EXP_%s[%d] = {%s};' % (qual, name, nw, ','.join('0x%016xull' % ((e >> (64 * i)) & (2 ** 64 - 1)) for i in range(nw))))

    fp('NBLS_P', P, raw=True)
    fp('NBLS_2P', 2 * P, raw=True)
    fp('NBLS_R1', 1)            # Montgomery one
    fp('NBLS_R2', RM)           # R^2 mod p, as mont(R)
    fp('NBLS_RAW_ONE', 1, raw=True)
    fp('NBLS_R3', RM * RM)                  # R^3 mod p as mont(R^2): montmul(t, R3) = t R^2
    fp('NBLS_TOP384', (1 << 384) * RM)     # montmul(t, TOP384) = t * 2^384 * R : high part of a 64-byte integer (hash_to_field)
    fp('NBLS_HALF_P_RAW', (P - 1) // 2, raw=True)   # v > (p-1)/2  <=>  floor(2v/p) = 1 (sign flags, index.ts:314)
    fp('NBLS_MASK381_RAW', (1 << 381) - 1, raw=True)
    fp('NBLS_POW2_381_RAW', 1 << 381, raw=True)
    fp('NBLS_POW2_383_RAW', 1 << 383, raw=True)
    a('#define NBLS_LIMB_BITS %d\n#define NBLS_NLIMBS %d\n#define NBLS_RBITS %d' % (bits, n, rbits))
    a('#define NBLS_N0_LIMB 0x%xu' % ((-pow(P, -1, 1 << bits)) % (1 << bits)))
    a('#define NBLS_N0_64 0x%016xull' % N0_64)
    a('#define NBLS_N0_32 0x%08xu' % N0_32)
    a('#define NBLS_X 0x%016xull' % X)
    fp('NBLS_INV3', pow(3, -1, P))   # 1/3 (folding a post-added term into a dot product that carries a multiplier 3)
    fp('NBLS_HALF', (P + 1) // 2)   # 1/2, used where the reference writes .div(2n) (math.ts:1349-1350)
    fp('NBLS_BETA', BETA)
    fp('NBLS_PSI2_C1', PSI2_C1[0])
    fp2('NBLS_PSI_X', PSI_X)
    fp2('NBLS_PSI_Y', PSI_Y)
    fp('NBLS_G1X', G1X)
    fp('NBLS_G1X_RAW', G1X, raw=True)
    fp('NBLS_NEG_G1Y_RAW', P - G1Y, raw=True)
    fp('NBLS_G1Y_RAW', G1Y, raw=True)
    fp('NBLS_G1Y', G1Y)
    fp2('NBLS_G2X', G2X)
    fp2('NBLS_G2Y', G2Y)
    fp2arr('NBLS_FROB12', FROB12)
    fp2arr('NBLS_FROB6_1', FROB6_1)
    fp2arr('NBLS_FROB6_2', FROB6_2)
    fp2arr('NBLS_ROOTS8', ROOTS8)
    fp2arr('NBLS_ROOTS8_INV', [f2inv(r) for r in ROOTS8[:4]])   # 1/R[k], k = 0..3 (Fp2.sqrt: candidate / root, math.ts:501)
    fp2arr('NBLS_ETAS', ETAS)
    fp2('NBLS_SWU_Z', SWU_Z)
    fp2('NBLS_SWU_A', SWU_A)
    fp2('NBLS_SWU_B', SWU_B)
    fp2arr('NBLS_ISO_XNUM', ISO_XNUM)
    fp2arr('NBLS_ISO_XDEN', ISO_XDEN)
    fp2arr('NBLS_ISO_YNUM', ISO_YNUM)
    fp2arr('NBLS_ISO_YDEN', ISO_YDEN)
    for k, v in EXP.items():
        exp(k, v)
    rw = [(R_ORDER >> (64 * i)) & (2 ** 64 - 1) for i in range(4)]
    a('%s uint64_t NBLS_R_ORDER[4] = {%s};' % (qual, ','.join('0x%016xull' % x for x in rw)))
    a('#endif')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        f.write('\n'.join(L) + '\n')
    print('wrote', path)


def main():
    emit(os.path.join(ROOT, 'oracle', 'consts_gen.h'), 64, 'fp', 'NBLS_ORACLE_CONSTS_GEN_H', 'static const')
    # HIP engine: 14 limbs of 28 bits (stored in u32), Montgomery radix R = 2^392 -- see DESIGN.md 3.1
    emit(os.path.join(ROOT, 'noble-bls12-381_amd', 'csrc', 'consts_gen.h'), 28, 'fp', 'NBLS_CONSTS_GEN_H', 'static const', rbits=392)


if __name__ == '__main__':
    main()
