"""ctypes binding of noble-bls12-381_amd/libnbls_sim.so: host simulator of the wave VM (test infrastructure)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'noble-bls12-381_amd')
RAW = 64          # bytes of one raw field element in scratch (14 limbs of 28 bits + padding)
F12 = 12 * RAW     # raw Fp12
P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def raw_elem(v):
    """standard integer -> raw Montgomery element (R = 2^392, 14 x 28-bit limbs in u32 words, 64 bytes)"""
    m = (v << 392) % P_MOD
    return b''.join(((m >> (28 * i)) & 0xfffffff).to_bytes(4, 'little') for i in range(14)) + bytes(8)


PROGS = ['MILLER_BYTES', 'MILLER_RAW', 'MILLER_FE', 'NORM_RAW', 'NORM_BYTES', 'FE_EASY', 'EXPX', 'FE_MID1', 'FE_MID2', 'FE_FINAL', 'MUL2', 'RAW_TO_BYTES', 'G1_VALIDATE', 'G2_VALIDATE', 'G1_DEC_A', 'G1_DEC_B', 'G2_DEC_A', 'G2_DEC_B', 'H2C_A', 'H2C_B',
         'G1_TO_PROJ', 'G1_ADD2', 'G1_NORM', 'G1_TO_AFFINE', 'G2_TO_PROJ', 'G2_ADD2', 'G2_NORM', 'G2_TO_AFFINE', 'T_SWU', 'T_ISO', 'T_CLEAR', 'H2C_C', 'MILLER_RAW2', 'G1_COMPRESS', 'G2_COMPRESS', 'H2C1_A', 'ENC1_A', 'H2C1_B', 'ENC1_B', 'G1_CLEAR', 'ENC2_A', 'ENC2_B', 'G1_MUL', 'G2_MUL', 'G1_ADD_AB', 'G2_ADD_AB', 'G1_HORNER', 'G2_HORNER', 'G1_SHIFTADD', 'G2_SHIFTADD', 'G1_MSM_PREP', 'G2_MSM_PREP', 'LINES_PQ', 'LINES_Q', 'LINES_BYTES', 'LINES_FROM_BYTES', 'ACC_BYTES', 'ACC_RAW', 'ACC_FE', 'ACC2_RAW', 'ACC_Q', 'G2_DEC_A192', 'G2_DEC_B192', 'G2_DEC_B_HEX', 'G1_FROM_RAW', 'G2_FROM_RAW', 'G2_SWAP', 'H2C_C1', 'H2C_C2', 'ACC4_RAW', 'MILLER_BYTES_LS', 'MILLER_RAW_LS', 'MILLER_FE_LS', 'EXPX_LS', 'EXPC_SQ', 'EXPC_DEC_A', 'EXPC_DEC_B', 'ACC8_RAW', 'H2C_C0', 'H2C_B1', 'H2C_B2', 'G1_MUL_W3', 'G2_MUL_W3', 'MUL2S', 'G1_MUL_FIXED', 'G2_MUL_GLS', 'MILLER_BYTES_LS2', 'MILLER_RAW_LS2', 'MILLER_FE_LS2', 'EXPX_LS2', 'G2_MUL_SAC', 'H2C_C1_LS2', 'H2C_C2_LS2', 'G2_MUL_SAC_LS2', 'H2C_NA', 'H2C_NM', 'H2C_NB']
P = {n: i for i, n in enumerate(PROGS)}


def load():
    if not os.environ.get('NBLS_SIM_NO_REBUILD'):   # set when a hand-built variant of the simulator is being checked
        subprocess.check_call(['make', '-s', '-C', os.path.join(PKG, 'csrc'), '../libnbls_sim.so'])
    return C.CDLL(os.path.join(PKG, 'libnbls_sim.so'))


def run(lib, prog, n, bufs):
    """bufs: dict index -> (ctypes buffer, stride)"""
    ptrs = (C.c_void_p * 8)()
    strides = (C.c_uint64 * 8)()
    for k, (b, s) in bufs.items():
        ptrs[k] = C.cast(b, C.c_void_p)
        strides[k] = s
    r = lib.nbls_sim_run(P[prog], C.c_uint(n), ptrs, strides)
    assert r == 0


EXPC_SQ_BYTES, EXPC_DEC_BYTES = 24 * RAW, 19 * RAW


def expx_compressed(lib, n, src, dst):
    """the compressed branch of expx() in csrc/pipelines_pairing.cpp on the simulator: Karabina's compressed squarings (EXPC_SQ), decompression around one inversion
    (EXPC_DEC_A, fp_inv, EXPC_DEC_B), flagged items (a vanishing g2) recomputed by the plain program.  Returns the flags."""
    KS = C.create_string_buffer(EXPC_SQ_BYTES * n); KD = C.create_string_buffer(EXPC_DEC_BYTES * n)
    KN = C.create_string_buffer(RAW * n); KNI = C.create_string_buffer(RAW * n); st = C.create_string_buffer(n)
    run(lib, 'EXPC_SQ', n, {3: (src, F12), 5: (KS, EXPC_SQ_BYTES)})
    run(lib, 'EXPC_DEC_A', n, {3: (KS, EXPC_SQ_BYTES), 4: (KN, RAW), 5: (KD, EXPC_DEC_BYTES)})
    lib.nbls_sim_fp_inv(C.c_uint(n), KN, KNI)
    run(lib, 'EXPC_DEC_B', n, {3: (KS, EXPC_SQ_BYTES), 4: (KNI, RAW), 6: (KD, EXPC_DEC_BYTES), 5: (dst, F12), 7: (st, 1)})
    flagged = [i for i in range(n) if st.raw[i]]
    for i in flagged:        # the device runs P_EXPX over an index list of the flagged items, in place
        a = C.create_string_buffer(src.raw[F12 * i:F12 * (i + 1)], F12); b = C.create_string_buffer(F12)
        run(lib, 'EXPX', 1, {3: (a, F12), 5: (b, F12)})
        C.memmove(C.addressof(dst) + F12 * i, b, F12)
    return flagged


def final_exp(lib, n, F, N, out, expx='EXPX'):
    """The launch sequence of final_exp_pipeline() in csrc/pipelines_pairing.cpp, on the simulator (expx='EXPC': the compressed-squaring form of the five exponentiations)."""
    NI = C.create_string_buffer(RAW * n)
    T = [C.create_string_buffer(F12 * n) for _ in range(7)]
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, 'FE_EASY', n, {3: (F, F12), 4: (NI, RAW), 5: (T[0], F12)})
    if expx == 'EXPC':
        ex = lambda a, b: expx_compressed(lib, n, a, b)
    else:
        ex = lambda a, b: run(lib, expx, n, {3: (a, F12), 5: (b, F12)})
    ex(T[0], T[1])
    run(lib, 'FE_MID1', n, {3: (T[0], F12), 5: (T[1], F12), 6: (T[2], F12)})
    ex(T[2], T[3])
    ex(T[3], T[4])
    ex(T[4], T[6])
    run(lib, 'FE_MID2', n, {3: (T[6], F12), 5: (T[1], F12), 6: (T[5], F12)})
    ex(T[5], T[6])
    bufs = {i: (T[i], F12) for i in range(7)}
    bufs[7] = (out, 576)
    run(lib, 'FE_FINAL', n, bufs)


def buf(data_or_size):
    if isinstance(data_or_size, int):
        return C.create_string_buffer(max(data_or_size, 1))
    return C.create_string_buffer(data_or_size, len(data_or_size))


def g1_decompress(lib, comp):
    n = len(comp) // 48
    X, R, Cd, out, st = buf(RAW * n), buf(RAW * n), buf(RAW * n), buf(96 * n), buf(n)
    inb = buf(comp)
    run(lib, 'G1_DEC_A', n, {0: (inb, 48), 3: (X, RAW), 4: (R, RAW)})
    lib.nbls_sim_fp_pow(C.c_uint(n), R, Cd, 0)
    run(lib, 'G1_DEC_B', n, {0: (inb, 48), 3: (X, RAW), 4: (R, RAW), 5: (Cd, RAW), 6: (out, 96), 7: (st, 1)})
    return out.raw, list(st.raw)


def g2_decompress(lib, comp, mode='sig'):
    """mode 'sig': PointG2.fromSignature on 96 bytes; 'sig192': its 192-byte branch; 'hex': PointG2.fromHex on 96 compressed bytes"""
    e = 192 if mode == 'sig192' else 96
    n = len(comp) // e
    X, R, Cd, out, st = buf(2 * RAW * n), buf(2 * RAW * n), buf(2 * RAW * n), buf(192 * n), buf(n)
    inb = buf(comp)
    run(lib, 'G2_DEC_A192' if mode == 'sig192' else 'G2_DEC_A', n, {0: (inb, e), 3: (X, 2 * RAW), 4: (R, 2 * RAW)})
    lib.nbls_sim_fp_pow(C.c_uint(n), R, Cd, 1)
    run(lib, {'sig': 'G2_DEC_B', 'sig192': 'G2_DEC_B192', 'hex': 'G2_DEC_B_HEX'}[mode], n, {0: (inb, e), 3: (X, 2 * RAW), 4: (R, 2 * RAW), 5: (Cd, 2 * RAW), 6: (out, 192), 7: (st, 1)})
    return out.raw, list(st.raw)


def hash_to_g2(lib, uniform, ls2=False, norm=False):
    """uniform: n * 256 bytes of expand_message_xmd output; ls2: the two ladders of clearCofactor in their two-lane forms (launches of at most 4096 messages);
    norm: the SWU square root by the norm method (two Fp exponentiations; launches of NBLS_H2C_NORM_MIN messages and more)"""
    n = len(uniform) // 256
    T, E, Pw, Q, N, NI, out, st = buf(4 * RAW * n), buf(4 * RAW * n), buf(4 * RAW * n), buf(6 * RAW * n), buf(RAW * n), buf(RAW * n), buf(192 * n), buf(n)
    St, Pt2 = buf(32 * RAW * n), buf(12 * RAW * n)
    E2 = buf(6 * RAW * n)
    # one SWU map per item (2 n items), then the two points of a message -> their sum on E2 (the launch sequence of dev_hash_to_g2 in csrc/pipelines_codec.cpp)
    if norm:
        run(lib, 'H2C_NA', n, {0: (buf(uniform), 256), 3: (T, 4 * RAW), 4: (E, 2 * RAW), 5: (St, 32 * RAW)})
        lib.nbls_sim_fp_pow(C.c_uint(2 * n), E, Pw, 0)
        St9 = (C.c_char * (32 * RAW * n - 9 * RAW)).from_buffer(St, 9 * RAW)
        run(lib, 'H2C_NM', 2 * n, {3: (T, 2 * RAW), 4: (St, 16 * RAW), 5: (Pw, RAW), 6: (St9, 16 * RAW), 7: (E, RAW)})
        lib.nbls_sim_fp_pow(C.c_uint(2 * n), E, Pw, 3)
        run(lib, 'H2C_NB', 2 * n, {3: (T, 2 * RAW), 4: (St, 16 * RAW), 5: (Pw, RAW), 6: (Pt2, 6 * RAW)})
    else:
        run(lib, 'H2C_A', n, {0: (buf(uniform), 256), 3: (T, 4 * RAW), 4: (E, 4 * RAW), 5: (St, 24 * RAW)})
        lib.nbls_sim_fp_pow(C.c_uint(2 * n), E, Pw, 2)
        run(lib, 'H2C_B1', 2 * n, {3: (T, 2 * RAW), 5: (Pw, 2 * RAW), 4: (St, 12 * RAW), 6: (Pt2, 6 * RAW)})
    run(lib, 'H2C_B2', n, {3: (Pt2, 12 * RAW), 6: (E2, 6 * RAW)})
    # clearCofactor in three programs: the points that do not depend on [x]P, then one program around each multiplication by x (the launch sequence of dev_clear_g2
    # in csrc/pipelines_codec.cpp, in place like there: C1 overwrites P with t1, C2 writes the result over t1)
    S = buf(6 * RAW * n)
    run(lib, 'H2C_C0', n, {3: (E2, 6 * RAW), 6: (Q, 6 * RAW), 5: (S, 6 * RAW)})
    run(lib, 'H2C_C1_LS2' if ls2 else 'H2C_C1', n, {3: (E2, 6 * RAW), 6: (Q, 6 * RAW)})
    run(lib, 'H2C_C2_LS2' if ls2 else 'H2C_C2', n, {3: (Q, 6 * RAW), 4: (E2, 6 * RAW), 5: (S, 6 * RAW), 6: (E2, 6 * RAW), 7: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, 'G2_TO_AFFINE', n, {3: (E2, 6 * RAW), 4: (NI, RAW), 2: (out, 192), 7: (st, 1)})
    return out.raw


def point_sum(lib, pts, g2=False):
    sz, psz = (192, 6 * RAW) if g2 else (96, 3 * RAW)
    pre = 'G2' if g2 else 'G1'
    n = len(pts) // sz
    A, Bf = buf(psz * (n + 2)), buf(psz * (n + 2))
    run(lib, pre + '_TO_PROJ', n, {(1 if g2 else 0): (buf(pts), sz), 3: (A, psz)})
    ident = bytearray(psz)
    yo = (2 * RAW) if g2 else RAW          # (0 : 1 : 0): y = Montgomery one
    ident[yo:yo + RAW] = raw_elem(1)
    m = n
    src, dst = A, Bf
    while m > 1:
        if m & 1:
            C.memmove(C.addressof(src) + m * psz, bytes(ident), psz)
            m += 1
        run(lib, pre + '_ADD2', m // 2, {3: (src, 2 * psz), 5: (dst, psz)})
        src, dst = dst, src
        m //= 2
    N, NI, out, st = buf(RAW), buf(RAW), buf(sz), buf(1)
    run(lib, pre + '_NORM', 1, {3: (src, psz), 4: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(1), N, NI)
    run(lib, pre + '_TO_AFFINE', 1, {3: (src, psz), 4: (NI, RAW), 2: (out, sz), 7: (st, 1)})
    return out.raw, st.raw[0]


def point_mul(lib, pts, scalars32, g2=False, w3=False):
    """dev_point_mul() of csrc/pipelines_codec.cpp on the simulator: ladder (2-bit windows; w3: the 3-bit form of small launches) -> inversion -> affine"""
    sz, psz = (192, 6 * RAW) if g2 else (96, 3 * RAW)
    pre = 'G2' if g2 else 'G1'
    n = len(scalars32) // 32
    Pj, N, NI, out, st = buf(psz * n), buf(RAW * n), buf(RAW * n), buf(sz * n), buf(n)
    run(lib, pre + ('_MUL_W3' if w3 else '_MUL'), n, {(1 if g2 else 0): (buf(pts), sz), 2: (buf(scalars32), 32), 3: (Pj, psz), 4: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, pre + '_TO_AFFINE', n, {3: (Pj, psz), 4: (NI, RAW), 2: (out, sz), 7: (st, 1)})
    return out.raw, st.raw


G1_FIXED_WIN = 3     # csrc/curve.h G1_FIXED_WIN (NBLS_G1FIXED_WIN)


def g1_fixed_table(lib, oracle):
    """ensure_g1_fixed() of csrc/pipelines_codec.cpp with the oracle in place of the device ladder: raw projective multiples [d 2^(WIN w)]G1, d = 1 .. 2^WIN - 1, per window"""
    win = G1_FIXED_WIN; nw = (256 + win - 1) // win; ne = (1 << win) - 1
    g = oracle.g1_generator()
    aff = b''
    for w in range(nw):
        for d in range(1, ne + 1):
            k = d << (win * w)
            aff += oracle.g1_mul(g, k if k < (1 << 256) else 1)[1]
    m = nw * ne
    tab = buf(3 * RAW * m)
    run(lib, 'G1_TO_PROJ', m, {0: (buf(aff), 96), 3: (tab, 3 * RAW)})
    return tab


def point_mul_fixed(lib, table, scalars32):
    """dev_point_mul() for the fixed base G1.BASE (getPublicKey): table sums (no doublings) -> inversion -> affine"""
    n = len(scalars32) // 32
    Pj, N, NI, out, st = buf(3 * RAW * n), buf(RAW * n), buf(RAW * n), buf(96 * n), buf(n)
    run(lib, 'G1_MUL_FIXED', n, {2: (buf(scalars32), 32), 5: (table, 0), 3: (Pj, 3 * RAW), 4: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, 'G1_TO_AFFINE', n, {3: (Pj, 3 * RAW), 4: (NI, RAW), 2: (out, 96), 7: (st, 1)})
    return out.raw, st.raw


def g2_affine_to_raw_projective(pts192, scale_seed=None):
    """n affine G2 wire points (x.c0 x.c1 y.c0 y.c1, 48 bytes big-endian each) -> raw projective points (six raw Montgomery elements each), the form sign's ladders take;
    scale_seed: multiply every point through by a pseudo-random z in Fp2 (another representative of the same point)"""
    import random
    rnd = random.Random(scale_seed)
    mul2 = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P_MOD, (a[0] * b[1] + a[1] * b[0]) % P_MOD)
    out = b''
    for i in range(len(pts192) // 192):
        w = [int.from_bytes(pts192[192 * i + 48 * k:192 * i + 48 * k + 48], 'big') for k in range(4)]
        x, y, z = (w[0], w[1]), (w[2], w[3]), (1, 0)
        if scale_seed is not None:
            z = (rnd.randrange(1, P_MOD), rnd.randrange(P_MOD))
            x, y = mul2(x, z), mul2(y, z)
        out += b''.join(raw_elem(v) for v in (x[0], x[1], y[0], y[1], z[0], z[1]))
    return out


def point_mul_gls(lib, pts192, scalars32):
    """dev_point_mul() for points known to lie in G2 (sign): base-|z| digits of the scalar (msm_decompose_kernel, here in Python) -> GLS ladder -> inversion -> affine"""
    Z = 0xd201000000010000
    n = len(scalars32) // 32
    dig = b''
    for i in range(n):
        k = int.from_bytes(scalars32[32 * i:32 * i + 32], 'big')
        a = []
        for _ in range(3):
            a.append(k % Z); k //= Z
        a.append(k)
        dig += b''.join(x.to_bytes(32, 'big') for x in a)
    psz = 6 * RAW
    Pj, N, NI, out, st = buf(psz * n), buf(RAW * n), buf(RAW * n), buf(192 * n), buf(n)
    run(lib, 'G2_MUL_GLS', n, {1: (buf(pts192), 192), 2: (buf(dig), 128), 3: (Pj, psz), 4: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, 'G2_TO_AFFINE', n, {3: (Pj, psz), 4: (NI, RAW), 2: (out, 192), 7: (st, 1)})
    return out.raw, st.raw


def sac_recode(k):
    """msm_sac_kernel of csrc/msm_kernels.hip restated: the four base-|z| digits of k recoded sign-aligned -> (S with the 'a0 was even' flag at bit 66, e1, e2, e3)"""
    Z = 0xd201000000010000
    a = []
    for _ in range(3):
        a.append(k % Z); k //= Z
    a.append(k)
    even = 1 - (a[0] & 1)
    a0 = a[0] | 1
    S = (a0 >> 1) | (1 << 65)
    out = [S | (even << 66)]
    for j in range(1, 4):
        v, e = a[j], 0
        for b in range(66):
            bit = v & 1
            e |= bit << b
            v = (v >> 1) + (bit & (1 - ((S >> b) & 1)))
        assert v == 0
        out.append(e)
    # the recoding represents the digits: a_j = sum_i s_i e_ji 2^i
    sgn = lambda i: 1 if (S >> i) & 1 else -1
    assert sum(sgn(i) << i for i in range(66)) == a0
    for j in range(1, 4):
        assert sum(sgn(i) * ((out[j] >> i) & 1) << i for i in range(66)) == a[j]
    return out


def point_mul_sac(lib, pts192, scalars32, prog='G2_MUL_SAC'):
    """dev_point_mul() for at most 6144 points known to lie in G2 (sign): sign-aligned recoding of the digits -> one-addition-per-bit ladder on raw PROJECTIVE base points (scaled representatives here;
    prog: its two-lane form up to 4096) -> inversion -> affine"""
    n = len(scalars32) // 32
    rc = b''.join(b''.join(x.to_bytes(32, 'big') for x in sac_recode(int.from_bytes(scalars32[32 * i:32 * i + 32], 'big'))) for i in range(n))
    psz = 6 * RAW
    Pj, N, NI, out, st = buf(psz * n), buf(RAW * n), buf(RAW * n), buf(192 * n), buf(n)
    run(lib, prog, n, {1: (buf(g2_affine_to_raw_projective(pts192, 98)), psz), 2: (buf(rc), 128), 3: (Pj, psz), 4: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, 'G2_TO_AFFINE', n, {3: (Pj, psz), 4: (NI, RAW), 2: (out, 192), 7: (st, 1)})
    return out.raw, st.raw


def msm(lib, pts, scalars32, nbits, g2=False):
    """dev_msm() of csrc/pipelines_codec.cpp on the simulator; the data-movement kernels of msm_kernels.hip (window digits, sort,
    gathers) are restated in Python, every group operation runs as a step program"""
    WB = 12
    sz, psz = (192, 6 * RAW) if g2 else (96, 3 * RAW)
    pre = 'G2' if g2 else 'G1'
    n = len(scalars32) // 32
    ident = bytearray(psz)
    yo = (2 * RAW) if g2 else RAW
    ident[yo:yo + RAW] = raw_elem(1)
    ident = bytes(ident)
    ks = [int.from_bytes(scalars32[32 * i:32 * i + 32], 'big') for i in range(n)]
    if nbits > 192:          # endomorphism split: k = sum a_i |z|^i, the points [|z|^i]P come from the PREP program
        Z = 0xd201000000010000
        dims = 4 if g2 else 2
        Pj = buf(psz * dims * max(n, 1))
        if n:
            run(lib, pre + '_MSM_PREP', n, {(1 if g2 else 0): (buf(pts), sz), 3: (Pj, dims * psz)})
        ks = [d for k in ks for d in ([k % Z, k // Z % Z, k // Z ** 2 % Z, k // Z ** 3] if g2 else [k % Z ** 2, k // Z ** 2])]
        n *= dims
        nbits = 65 if g2 else 129
    else:
        Pj = buf(psz * max(n, 1))
        if n:
            run(lib, pre + '_TO_PROJ', n, {(1 if g2 else 0): (buf(pts), sz), 3: (Pj, psz)})
    nwin = (nbits + WB - 1) // WB
    pairs = sorted(((w << WB) | ((ks[i] >> (WB * w)) & ((1 << WB) - 1)), w * n + i) for w in range(nwin) for i in range(n))   # stable: ties keep input order
    keys = [k for k, _ in pairs]
    m = len(pairs)
    P = bytearray(b''.join(Pj.raw[psz * (v % n):psz * (v % n) + psz] for _, v in pairs)) if m else bytearray()
    maxrun, j = 0, 0
    while j < m:
        e = j
        while e < m and keys[e] == keys[j]:
            e += 1
        maxrun = max(maxrun, e - j); j = e
    pos, j = [0] * m, 0
    while j < m:
        e = j
        while e < m and keys[e] == keys[j]:
            pos[e] = e - j; e += 1
        j = e
    d = 1
    while d < maxrun:          # balanced tree inside every run: rank multiple of 2d absorbs the element d further on
        lst = [j for j in range(m) if pos[j] % (2 * d) == 0 and j + d < m and keys[j + d] == keys[j]]
        assert len(lst) <= m // (d + 1) + 1
        A = buf(b''.join(bytes(P[psz * j:psz * j + psz]) for j in lst))
        Bv = buf(b''.join(bytes(P[psz * (j + d):psz * (j + d) + psz]) for j in lst))
        run(lib, pre + '_ADD_AB', len(lst), {3: (A, psz), 4: (Bv, psz), 5: (A, psz)})     # in place, as on the device
        for i, j in enumerate(lst):
            P[psz * j:psz * j + psz] = A.raw[psz * i:psz * i + psz]
        d *= 2
    buckets = [ident] * (nwin << WB)
    for j in range(m):
        if j == 0 or keys[j - 1] != keys[j]:
            buckets[keys[j]] = bytes(P[psz * j:psz * j + psz])
    G = []
    for w in range(nwin):
        for t in range(WB):
            for jj in range(1 << (WB - 1)):
                b = ((jj >> t) << (t + 1)) | (1 << t) | (jj & ((1 << t) - 1))
                G.append(buckets[(w << WB) + b])
    cnt = len(G)
    src = buf(b''.join(G))
    while cnt > nwin * WB:
        dst = buf(psz * (cnt // 2))
        run(lib, pre + '_ADD2', cnt // 2, {3: (src, 2 * psz), 5: (dst, psz)})
        src = dst
        cnt //= 2
    S = buf(psz * nwin)
    run(lib, pre + '_HORNER', nwin, {3: (src, WB * psz), 5: (S, psz)})
    acc = buf(S.raw[psz * (nwin - 1):psz * nwin])
    for w in range(nwin - 2, -1, -1):
        run(lib, pre + '_SHIFTADD', 1, {3: (acc, psz), 4: (buf(S.raw[psz * w:psz * w + psz]), psz), 5: (acc, psz)})
    N, NI, out, st = buf(RAW), buf(RAW), buf(sz), buf(1)
    run(lib, pre + '_NORM', 1, {3: (acc, psz), 4: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(1), N, NI)
    run(lib, pre + '_TO_AFFINE', 1, {3: (acc, psz), 4: (NI, RAW), 2: (out, sz), 7: (st, 1)})
    return out.raw, st.raw[0]


def compress(lib, aff, g2=False):
    sz = 192 if g2 else 96
    n = len(aff) // sz
    out = buf(n * sz // 2)
    run(lib, 'G2_COMPRESS' if g2 else 'G1_COMPRESS', n, {0: (buf(aff), sz), 2: (out, sz // 2)})
    return out.raw


def hash_to_g1(lib, uniform, count):
    """dev_hash_to_g1() of csrc/pipelines_codec.cpp on the simulator; uniform: n * 64 * count bytes of expand_message_xmd output"""
    n = len(uniform) // (64 * count)
    us = count * RAW
    U, E, Pw, Q, Q2, N, NI, out, st = buf(us * n), buf(us * n), buf(us * n), buf(3 * RAW * n), buf(3 * RAW * n), buf(RAW * n), buf(RAW * n), buf(96 * n), buf(n)
    run(lib, 'H2C1_A' if count == 2 else 'ENC1_A', n, {0: (buf(uniform), 64 * count), 3: (U, us), 4: (E, us)})
    lib.nbls_sim_fp_pow(C.c_uint(count * n), E, Pw, 3)
    run(lib, 'H2C1_B' if count == 2 else 'ENC1_B', n, {3: (U, us), 5: (Pw, us), 6: (Q, 3 * RAW)})
    run(lib, 'G1_CLEAR', n, {3: (Q, 3 * RAW), 6: (Q2, 3 * RAW), 7: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, 'G1_TO_AFFINE', n, {3: (Q2, 3 * RAW), 4: (NI, RAW), 2: (out, 96), 7: (st, 1)})
    return out.raw


def encode_to_g2(lib, uniform):
    n = len(uniform) // 128
    T, E, Pw, E2, Q, N, NI, out, st = buf(2 * RAW * n), buf(2 * RAW * n), buf(2 * RAW * n), buf(6 * RAW * n), buf(6 * RAW * n), buf(RAW * n), buf(RAW * n), buf(192 * n), buf(n)
    run(lib, 'ENC2_A', n, {0: (buf(uniform), 128), 3: (T, 2 * RAW), 4: (E, 2 * RAW)})
    lib.nbls_sim_fp_pow(C.c_uint(n), E, Pw, 2)
    run(lib, 'ENC2_B', n, {3: (T, 2 * RAW), 5: (Pw, 2 * RAW), 6: (E2, 6 * RAW)})
    run(lib, 'H2C_C', n, {3: (E2, 6 * RAW), 6: (Q, 6 * RAW), 7: (N, RAW)})
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, 'G2_TO_AFFINE', n, {3: (Q, 6 * RAW), 4: (NI, RAW), 2: (out, 192), 7: (st, 1)})
    return out.raw
