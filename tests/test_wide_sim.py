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
