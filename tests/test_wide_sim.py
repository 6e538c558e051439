"""The one-limb-per-lane interpreter (csrc/wide_exec.h; device: vm_wide_kernel.hip, one item per workgroup, a lane-op per row of sixteen lanes, two barriers per step) on the
host: the same compiled programs, executed with the row arithmetic the device runs -- lazily normalised signed limbs, the Montgomery reduction spread over the lanes -- and
checked bit-for-bit against the reference-generated vectors and the oracle through the same pipelines as tests/test_vm_sim.py.  The host model also counts every violated
32 / 64-bit assumption of the device code (nbls_sim_wide_violations)."""
import ctypes as C
import pytest
import vmsim_py
import test_vm_sim as T


@pytest.fixture(scope='module')
def sim():
    lib = vmsim_py.load()
    lib.nbls_sim_wide_violations.restype = C.c_ulong
    lib.nbls_sim_set_wide(1)
    lib.nbls_sim_wide_violations()
    yield lib
    lib.nbls_sim_set_wide(0)


def test_programs_the_form_implements(sim):
    ok = {n for n in vmsim_py.P if sim.nbls_sim_wide_supported(vmsim_py.P[n])}
    # the final exponentiation between its first and last program, the product tree, the accumulation programs, the cofactor ladders
    assert {'EXPX', 'FE_MID1', 'FE_MID2', 'FE_EASY', 'NORM_RAW', 'MUL2', 'MUL2S', 'ACC_RAW', 'ACC_FE', 'H2C_C1', 'H2C_C2'} <= ok
    # lane-split forms, programs with wire loads / stores or halvings stay on the other forms
    assert not ({'EXPX_LS', 'EXPX_LS2', 'MILLER_FE', 'FE_FINAL', 'LINES_PQ'} & ok)


def test_pairing_pipelines_with_wide_programs(sim, oracle, golden, testdata):
    T.test_full_pairing_pipeline(sim, oracle, golden)
    T.test_final_exp_and_product(sim, oracle, golden, testdata)
    T.test_split_miller(sim, oracle, golden)
    assert sim.nbls_sim_wide_violations() == 0


def test_wide_expx_agrees_with_the_interpreter_on_extreme_elements(sim):
    """EXPX alone on extreme raw inputs: the wide form and the interpreter agree modulo p on every output (their representatives differ), outputs stay below the bound scratch
    elements are reloaded with, limbs are exact"""
    import random
    n = 6
    rnd = random.Random(11)
    p = vmsim_py.P_MOD
    vals = [[rnd.randrange(p) for _ in range(12)] for _ in range(n - 3)] + [[p - 1] * 12, [1] + [0] * 11, [7 * p + 5] * 12]
    src = C.create_string_buffer(b''.join(b''.join(vmsim_py.raw_elem(v) for v in item) for item in vals), vmsim_py.F12 * n)
    outs = []
    for wide in (1, 0):
        sim.nbls_sim_set_wide(wide)
        dst = C.create_string_buffer(vmsim_py.F12 * n)
        vmsim_py.run(sim, 'EXPX', n, {3: (src, vmsim_py.F12), 5: (dst, vmsim_py.F12)})
        outs.append(dst.raw)
    sim.nbls_sim_set_wide(1)
    words = lambda raw, k: [int.from_bytes(raw[64 * k + 4 * i:64 * k + 4 * i + 4], 'little') for i in range(16)]
    for k in range(12 * n):
        wa, wb = words(outs[0], k), words(outs[1], k)
        assert all(x < (1 << 28) for x in wa[:14]) and wa[14:] == [0, 0], k
        a = sum(x << (28 * i) for i, x in enumerate(wa[:14])); b = sum(x << (28 * i) for i, x in enumerate(wb[:14]))
        assert (a - b) % p == 0 and 0 <= a < 8 * p, k
    assert sim.nbls_sim_wide_violations() == 0


def test_g1_window_combination_with_one_limb_per_lane(sim):
    """csrc/g1_wide.h (device: nbls_g1_wide_combine_kernel, one wavefront): sum_w 2^(12 w) S_w over eleven projective window sums -- the tail of the G1 multi-scalar
    multiplication -- against plain affine arithmetic on y^2 = x^3 + 4 in Python, for ordinary points, a window sum that is the identity, equal and opposite neighbours (the
    formulas are the complete ones), raw inputs at the top of the range scratch elements may have; outputs exact and below 8 p; no 32 / 64-bit assumption violated."""
    import random
    p = vmsim_py.P_MOD; R = 1 << 392; RAW = vmsim_py.RAW
    rnd = random.Random(1212)

    def add(P1, P2):
        if P1 is None: return P2
        if P2 is None: return P1
        (x1, y1), (x2, y2) = P1, P2
        if x1 == x2:
            if (y1 + y2) % p == 0: return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def mul(k, P1):
        acc = None
        while k:
            if k & 1: acc = add(acc, P1)
            P1 = add(P1, P1); k >>= 1
        return acc

    def point():
        while True:
            x = rnd.randrange(p); y2 = (x * x * x + 4) % p; y = pow(y2, (p + 1) // 4, p)
            if y * y % p == y2: return (x, y)

    def raw_proj(P1, big=False):
        # a projective representative with a random Z; `big`: coordinates as non-canonical representatives up to 7.9 p (what a program may leave in scratch)
        if P1 is None: c = [0, rnd.randrange(1, p), 0]
        else:
            z = rnd.randrange(1, p); c = [P1[0] * z % p, P1[1] * z % p, z]
        vals = [(v * R) % p + (rnd.randrange(7) * p if big else 0) for v in c]
        assert all(v < (1 << 392) for v in vals)
        return b''.join(b''.join(((v >> (28 * i)) & 0xfffffff).to_bytes(4, 'little') for i in range(13)) + (v >> 364).to_bytes(4, 'little') + bytes(8) for v in vals)

    def unraw(buf):
        out = []
        for k in range(3):
            o = buf[64 * k:64 * k + 64]
            limbs = [int.from_bytes(o[4 * i:4 * i + 4], 'little') for i in range(16)]
            assert all(l < (1 << 28) for l in limbs[:13]) and limbs[14] == 0 and limbs[15] == 0
            v = sum(l << (28 * i) for i, l in enumerate(limbs[:14]))
            assert v < 8 * p
            out.append(v * pow(R, -1, p) % p)
        return out

    nwin, shift = 11, 12
    for trial in range(6):
        W = [point() for _ in range(nwin)]
        if trial == 1: W[3] = None; W[10] = None                 # identity window sums, also the top one
        if trial == 2: W[5] = W[6]; W[7] = (W[8][0], (-W[8][1]) % p)
        if trial == 3: W = [None] * nwin
        if trial == 4:                                           # the running sum meets its own negative: 2^12 acc + S = 0 at window 4
            acc = None
            for w in range(nwin - 1, 4, -1): acc = add(mul(1 << shift, acc), W[w]) if acc is not None else W[w]
            t = mul(1 << shift, acc); W[4] = (t[0], (-t[1]) % p)
        S = vmsim_py.buf(b''.join(raw_proj(Q, big=(trial == 5)) for Q in W))
        out = vmsim_py.buf(3 * RAW)
        sim.nbls_sim_g1_wide_combine(S, nwin, shift, out)
        X, Y, Z = unraw(out.raw)
        exp = None
        for w in range(nwin - 1, -1, -1): exp = add(mul(1 << shift, exp), W[w])
        if exp is None:
            assert X == 0 and Z == 0 and Y != 0, trial
        else:
            assert Z != 0 and X * pow(Z, -1, p) % p == exp[0] and Y * pow(Z, -1, p) % p == exp[1], trial
    assert sim.nbls_sim_wide_violations() == 0
