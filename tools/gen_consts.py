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

# ---- G1 hash-to-curve (RFC 9380 section 8.8.1, appendix E.2; reference math.ts:1270-1313, 1612-1790; index.ts:331-350)
# SWU on the 11-isogenous curve E1': y^2 = x^3 + A1' x + B1', Z = 11; 11-isogeny E1' -> E1 (coefficients highest degree first)
G1_SWU_A = 0x144698a3b8e9433d693a02c96d4982b0ea985383ee66a8d8e8981aefd881ac98936f8da0e0f97f5cf428082d584c1d
G1_SWU_B = 0x12e2908d11688030018b12e8753eee3b2016c1f0f24f4070a0b9c14fcef35ef55a23215a316ceaa5d1cc48e98e172be0
G1_SWU_Z = 11
G1_ISO_XNUM = [
    0x06e08c248e260e70bd1e962381edee3d31d79d7e22c837bc23c0bf1bc24c6b68c24b1b80b64d391fa9c8ba2e8ba2d229,
    0x10321da079ce07e272d8ec09d2565b0dfa7dccdde6787f96d50af36003b14866f69b771f8c285decca67df3f1605fb7b,
    0x169b1f8e1bcfa7c42e0c37515d138f22dd2ecb803a0c5c99676314baf4bb1b7fa3190b2edc0327797f241067be390c9e,
    0x080d3cf1f9a78fc47b90b33563be990dc43b756ce79f5574a2c596c928c5d1de4fa295f296b74e956d71986a8497e317,
    0x17b81e7701abdbe2e8743884d1117e53356de5ab275b4db1a682c62ef0f2753339b7c8f8c8f475af9ccb5618e3f0c88e,
    0x0d6ed6553fe44d296a3726c38ae652bfb11586264f0f8ce19008e218f9c86b2a8da25128c1052ecaddd7f225a139ed84,
    0x1630c3250d7313ff01d1201bf7a74ab5db3cb17dd952799b9ed3ab9097e68f90a0870d2dcae73d19cd13c1c66f652983,
    0x0e99726a3199f4436642b4b3e4118e5499db995a1257fb3f086eeb65982fac18985a286f301e77c451154ce9ac8895d9,
    0x1778e7166fcc6db74e0609d307e55412d7f5e4656a8dbf25f1b33289f1b330835336e25ce3107193c5b388641d9b6861,
    0x0d54005db97678ec1d1048c5d10a9a1bce032473295983e56878e501ec68e25c958c3e3d2a09729fe0179f9dac9edcb0,
    0x17294ed3e943ab2f0588bab22147a81c7c17e75b2f6a8417f565e33c70d1e86b4838f2a6f318c356e834eef1b3cb83bb,
    0x11a05f2b1e833340b809101dd99815856b303e88a2d7005ff2627b56cdb4e2c85610c2d5f2e62d6eaeac1662734649b7,
]
G1_ISO_XDEN = [
    0x000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000001,
    0x095fc13ab9e92ad4476d6e3eb3a56680f682b4ee96f7d03776df533978f31c1593174e4b4b7865002d6384d168ecdd0a,
    0x0a10ecf6ada54f825e920b3dafc7a3cce07f8d1d7161366b74100da67f39883503826692abba43704776ec3a79a1d641,
    0x14a7ac2a9d64a8b230b3f5b074cf01996e7f63c21bca68a81996e1cdf9822c580fa5b9489d11e2d311f7d99bbdcc5a5e,
    0x0772caacf16936190f3e0c63e0596721570f5799af53a1894e2e073062aede9cea73b3538f0de06cec2574496ee84a3a,
    0x0e7355f8e4e667b955390f7f0506c6e9395735e9ce9cad4d0a43bcef24b8982f7400d24bc4228f11c02df9a29f6304a5,
    0x13a8e162022914a80a6f1d5f43e7a07dffdfc759a12062bb8d6b44e833b306da9bd29ba81f35781d539d395b3532a21e,
    0x03425581a58ae2fec83aafef7c40eb545b08243f16b1655154cca8abc28d6fd04976d5243eecf5c4130de8938dc62cd8,
    0x0b2962fe57a3225e8137e629bff2991f6f89416f5a718cd1fca64e00b11aceacd6a3d0967c94fedcfcc239ba5cb83e19,
    0x12561a5deb559c4348b4711298e536367041e8ca0cf0800c0126c2588c48bf5713daa8846cb026e9e5c8276ec82b3bff,
    0x08ca8d548cff19ae18b2e62f4bd3fa6f01d5ef4ba35b48ba9c9588617fc8ac62b558d681be343df8993cf9fa40d21b1c,
]
G1_ISO_YNUM = [
    0x15e6be4e990f03ce4ea50b3b42df2eb5cb181d8f84965a3957add4fa95af01b2b665027efec01c7704b456be69c8b604,
    0x05c129645e44cf1102a159f748c4a3fc5e673d81d7e86568d9ab0f5d396a7ce46ba1049b6579afb7866b1e715475224b,
    0x0245a394ad1eca9b72fc00ae7be315dc757b3b080d4c158013e6632d3c40659cc6cf90ad1c232a6442d9d3f5db980133,
    0x0b182cac101b9399d155096004f53f447aa7b12a3426b08ec02710e807b4633f06c851c1919211f20d4c04f00b971ef8,
    0x18b46a908f36f6deb918c143fed2edcc523559b8aaf0c2462e6bfe7f911f643249d9cdf41b44d606ce07c8a4d0074d8e,
    0x19713e47937cd1be0dfd0b8f1d43fb93cd2fcbcb6caf493fd1183e416389e61031bf3a5cce3fbafce813711ad011c132,
    0x0e1bba7a1186bdb5223abde7ada14a23c42a0ca7915af6fe06985e7ed1e4d43b9b3f7055dd4eba6f2bafaaebca731c30,
    0x09fc4018bd96684be88c9e221e4da1bb8f3abd16679dc26c1e8b6e6a1f20cabe69d65201c78607a360370e577bdba587,
    0x0987c8d5333ab86fde9926bd2ca6c674170a05bfe3bdd81ffd038da6c26c842642f64550fedfe935a15e4ca31870fb29,
    0x04ab0b9bcfac1bbcb2c977d027796b3ce75bb8ca2be184cb5231413c4d634f3747a87ac2460f415ec961f8855fe9d6f2,
    0x16603fca40634b6a2211e11db8f0a6a074a7d0d4afadb7bd76505c3d3ad5544e203f6326c95a807299b23ab13633a5f0,
    0x08cc03fdefe0ff135caf4fe2a21529c4195536fbe3ce50b879833fd221351adc2ee7f8dc099040a841b6daecf2e8fedb,
    0x01f86376e8981c217898751ad8746757d42aa7b90eeb791c09e4a3ec03251cf9de405aba9ec61deca6355c77b0e5f4cb,
    0x00cc786baa966e66f4a384c86a3b49942552e2d658a31ce2c344be4b91400da7d26d521628b00523b8dfe240c72de1f6,
    0x134996a104ee5811d51036d776fb46831223e96c254f383d0f906343eb67ad34d6c56711962fa8bfe097e75a2e41c696,
    0x090d97c81ba24ee0259d1f094980dcfa11ad138e48a869522b52af6c956543d3cd0c7aee9b3ba3c2be9845719707bb33,
]
G1_ISO_YDEN = [
    0x000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000001,
    0x0e0fa1d816ddc03e6b24255e0d7819c171c40f65e273b853324efcd6356caa205ca2f570f13497804415473a1d634b8f,
    0x02660400eb2e4f3b628bdd0d53cd76f2bf565b94e72927c1cb748df27942480e420517bd8714cc80d1fadc1326ed06f7,
    0x0ad6b9514c767fe3c3613144b45f1496543346d98adf02267d5ceef9a00d9b8693000763e3b90ac11e99b138573345cc,
    0x0accbb67481d033ff5852c1e48c50c477f94ff8aefce42d28c0f9a88cea7913516f968986f7ebbea9684b529e2561092,
    0x04d2f259eea405bd48f010a01ad2911d9c6dd039bb61a6290e591b36e636a5c871a5c29f4f83060400f8b49cba8f6aa8,
    0x167a55cda70a6e1cea820597d94a84903216f763e13d87bb5308592e7ea7d4fbc7385ea3d529b35e346ef48bb8913f55,
    0x1866c8ed336c61231a1be54fd1d74cc4f9fb0ce4c6af5920abc5750c4bf39b4852cfe2f7bb9248836b233d9d55535d4a,
    0x16a3ef08be3ea7ea03bcddfabba6ff6ee5a4375efa1f4fd7feb34fd206357132b920f5b00801dee460ee415a15812ed9,
    0x166007c08a99db2fc3ba8734ace9824b5eecfdfa8d0cf8ef5dd365bc400a0051d5fa9c01a58b1fb93d1a1399126a775c,
    0x08d9e5297186db2d9fb266eaac783182b70152c65550d881c5ecd87b6f0f5a6449f38db9dfa9cce202c6477faaf9b7ac,
    0x0be0e079545f43e4b00cc912f8228ddcc6d19c9f0f69bbb0542eda0fc9dec916a20b15dc0fd2ededda39142311a5001d,
    0x16b7d288798e5395f20d23bf89edb4d1d115c5dbddbcd30e123da489e726af41727364f2c28297ada8d26d98445f5416,
    0x058df3306640da276faaae7d6e8eb15778c4855551ae7f310c35a5dd279cd2eca6757cd636f96f891e2538b53dbf67f2,
    0x1962d75c2381201e1a0cbd6c43c348b885c84ff731c4d59ca4a10356f453e01f78a4260763529e3532f6102c2e49a03d,
    0x16112c4c3a9c98b252181140fad0eae9601a6de578980be6eec3232b5be72e7a07f3688ef60c206d01479253b03663c1,
]


def _fp_sqrt(a):
    r = pow(a, (P + 1) // 4, P)
    return r if r * r % P == a % P else None


G1_SWU_C2 = _fp_sqrt(pow(P - G1_SWU_Z, 3, P))      # sqrt((-Z)^3), math.ts:1281 (the root Fp.sqrt returns: a^((p+1)/4))
assert G1_SWU_C2 is not None
# hash-to-G2 by the norm method (codec.h swu_norm_*): when u / v is not a square the root of Z^3 t^6 u / v is taken, and the root of ITS norm is sqrt(-N(Z)^3) N(t)^3 n with
# n^2 = -N(u conj(v)) -- N(Z) = 5 is a non-residue (Z is a non-square of Fp2), so -125 is a square of Fp
SWU_SQRT_M125 = _fp_sqrt(P - 125)
assert SWU_SQRT_M125 is not None and (SWU_Z[0] ** 2 + SWU_Z[1] ** 2) % P == 5


def _g1_iso_check():
    """a point of E1' must land on E1: y^2 = x^3 + 4"""
    def horner(cs, x):
        acc = cs[0]
        for c in cs[1:]:
            acc = (acc * x + c) % P
        return acc
    t = 2
    while True:
        rhs = (t * t * t + G1_SWU_A * t + G1_SWU_B) % P
        y = _fp_sqrt(rhs)
        if y is not None:
            break
        t += 1
    xn, xd, yn, yd = (horner(c, t) for c in (G1_ISO_XNUM, G1_ISO_XDEN, G1_ISO_YNUM, G1_ISO_YDEN))
    X = xn * pow(xd, -1, P) % P
    Y = y * yn % P * pow(yd, -1, P) % P
    assert (Y * Y - X * X * X - 4) % P == 0, "G1 isogeny constants do not map E1' to E1"
    assert [len(c) for c in (G1_ISO_XNUM, G1_ISO_XDEN, G1_ISO_YNUM, G1_ISO_YDEN)] == [12, 11, 16, 16] and G1_ISO_XDEN[0] == 1 and G1_ISO_YDEN[0] == 1


_g1_iso_check()

# exponents
EXP = {
    'P_MINUS_2': P - 2,
    'P_PLUS_1_DIV_4': (P + 1) // 4,
    'P2_PLUS_7_DIV_16': (P * P + 7) // 16,      # Fp2.sqrt candidate exponent (math.ts:493, ORDER = p^2 - 1)
    'P2_MINUS_9_DIV_16': (P * P - 9) // 16,     # sqrt_div_fp2 (math.ts:1191)
    'P_MINUS_3_DIV_4': (P - 3) // 4,            # map_to_curve_simple_swu_3mod4 (math.ts:1279)
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

    def fparr(name, vs):
        a('%s %s %s[%d][%d] = {' % (qual, word, name, len(vs), n))
        for v in vs:
            a('  %s,' % c_fp(v, bits, False, rbits))
        a('};')

    def exp(name, e):
        nw = (e.bit_length() + 63) // 64
        a('#define NBLS_%s_BITS %d' % (name, e.bit_length()))
        a('%s uint64_t NBLS_EXP_%s[%d] = {%s};' % (qual, name, nw, ','.join('0x%016xull' % ((e >> (64 * i)) & (2 ** 64 - 1)) for i in range(nw))))

    def joint_digits(name, e):
        # e = c0 + c1 * p: a^e = a^c0 * conj(a)^c1 in Fp2 (a^p = conj(a)); joint 2-bit windows, most significant first, digit = c1_bits << 2 | c0_bits
        c0, c1 = e % P, e // P
        assert c0 + c1 * P == e and c1 < P
        nwin = (max(c0.bit_length(), c1.bit_length()) + 1) // 2
        ds = [(((c1 >> (2 * w)) & 3) << 2) | ((c0 >> (2 * w)) & 3) for w in reversed(range(nwin))]
        a('%s unsigned char NBLS_JOINT_%s[%d] = {%s};' % (qual, name, nwin, ','.join(str(d) for d in ds)))

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
    fp('NBLS_SWU_SQRT_M125', SWU_SQRT_M125)
    fp2arr('NBLS_ISO_XNUM', ISO_XNUM)
    fp2arr('NBLS_ISO_XDEN', ISO_XDEN)
    fp2arr('NBLS_ISO_YNUM', ISO_YNUM)
    fp2arr('NBLS_ISO_YDEN', ISO_YDEN)
    fp('NBLS_G1_SWU_A', G1_SWU_A)
    fp('NBLS_G1_SWU_B', G1_SWU_B)
    fp('NBLS_G1_SWU_Z', G1_SWU_Z)
    fp('NBLS_G1_SWU_C2', G1_SWU_C2)
    fparr('NBLS_G1_ISO_XNUM', G1_ISO_XNUM)
    fparr('NBLS_G1_ISO_XDEN', G1_ISO_XDEN)
    fparr('NBLS_G1_ISO_YNUM', G1_ISO_YNUM)
    fparr('NBLS_G1_ISO_YDEN', G1_ISO_YDEN)
    for k, v in EXP.items():
        exp(k, v)
        if v > P:
            joint_digits(k, v)
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
