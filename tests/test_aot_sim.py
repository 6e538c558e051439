"""The translated programs of the ahead-of-time kernels (csrc/aot.h, aot_exec.h: absolute descriptors, K_DOT finish in the 64-bit columns) executed by the host
simulator and checked bit-for-bit against the reference-generated vectors and oracle/ (CPU only): the same pipelines as tests/test_vm_sim.py with
nbls_sim_set_aot(1), i.e. EXPX, ACC_FE, ACC4_RAW and LINES_PQ run through aot_translate + aot_step instead of the interpreter's semantics."""
import ctypes as C
import pytest
import vmsim_py
import test_vm_sim as T
from goldenio import hx


@pytest.fixture(scope='module')
def sim():
    lib = vmsim_py.load()
    lib.nbls_sim_set_aot(1)
    yield lib
    lib.nbls_sim_set_aot(0)


def test_listed_programs_translate(sim):
    for name in ('EXPX', 'ACC_FE', 'ACC4_RAW', 'LINES_PQ'):
        assert sim.nbls_sim_has_aot(vmsim_py.P[name]) == 1, name
    for name in ('EXPX_LS', 'MILLER_FE_LS', 'MILLER_RAW_LS', 'MILLER_BYTES_LS'):     # the lane-split forms have kernels of their own (aot.h NBLS_AOT_LS_KERNELS)
        assert sim.nbls_sim_has_aot(vmsim_py.P[name]) == 1, name
    assert sim.nbls_sim_has_aot(vmsim_py.P['EXPC_SQ']) == 0     # the compressed-squaring experiment stays on the interpreter


def test_final_exponentiation_through_translated_expx(sim, oracle, golden, testdata):
    T.test_full_pairing_pipeline(sim, oracle, golden)
    T.test_final_exp_and_product(sim, oracle, golden, testdata)


def test_split_miller_through_translated_programs(sim, oracle, golden):
    T.test_split_miller(sim, oracle, golden)


def test_translated_expx_on_extreme_elements(sim, oracle):
    """EXPX alone on unitary elements built from extreme inputs: f -> easy part (interpreter) -> x-th power (translated) against the oracle's final exponentiation
    is covered above; here the translated and the interpreted program must agree mod p on raw outputs of the same inputs (their weak reductions differ)."""
    n = 6
    import random
    rnd = random.Random(7)
    p = vmsim_py.P_MOD
    vals = [[rnd.randrange(p) for _ in range(12)] for _ in range(n - 2)] + [[p - 1] * 12, [1] + [0] * 11]
    src = C.create_string_buffer(b''.join(b''.join(vmsim_py.raw_elem(v) for v in item) for item in vals), vmsim_py.F12 * n)
    outs = []
    for aot in (1, 0):
        sim.nbls_sim_set_aot(aot)
        dst = C.create_string_buffer(vmsim_py.F12 * n)
        vmsim_py.run(sim, 'EXPX', n, {3: (src, vmsim_py.F12), 5: (dst, vmsim_py.F12)})
        outs.append(dst.raw)
    sim.nbls_sim_set_aot(1)
    def elems(raw):
        return [sum(int.from_bytes(raw[64 * k + 4 * i:64 * k + 4 * i + 4], 'little') << (28 * i) for i in range(14)) for k in range(len(raw) // 64)]
    a, b = elems(outs[0]), elems(outs[1])
    assert all((x - y) % p == 0 for x, y in zip(a, b))
    assert all(0 <= x < 8 * p for x in a)     # scratch elements are reloaded with bound 8 (trace.h outputw)


def test_lane_split_programs_translated(sim, oracle, golden):
    """the lane-split forms through aot_translate + aot_step<..., LS = 4>: every sub-lane's own round addresses, the columns of the four sub-lanes summed before the one
    finish (on the device two DPP stages; the simulator visits the lanes from the top down and sums what the partners left), everything else on sub-lane 0"""
    T.test_lane_split_programs.__wrapped__(sim, oracle, golden, '_LS') if hasattr(T.test_lane_split_programs, '__wrapped__') else T.test_lane_split_programs(sim, oracle, golden, '_LS')
    for name in ('EXPX_LS2', 'MILLER_FE_LS2', 'MILLER_RAW_LS2', 'MILLER_BYTES_LS2'):       # round 5: the two-lane forms (one DPP stage) have kernels of their own
        assert sim.nbls_sim_has_aot(vmsim_py.P[name]) == 1, name
    T.test_lane_split_programs(sim, oracle, golden, '_LS2')


def test_decode_and_hash_programs_translated(sim, oracle, golden, testdata):
    """the programs with flags, selects and status steps (G1 decode + subgroup check, hash-to-G2 and its cofactor clearing) through the translated form"""
    for name in ('H2C_A', 'H2C_B1', 'H2C_B2', 'H2C_C0', 'H2C_C1', 'H2C_C2', 'G1_DEC_A', 'G1_DEC_B', 'G2_TO_AFFINE', 'MUL2', 'FE_MID1', 'FE_MID2'):
        assert sim.nbls_sim_has_aot(vmsim_py.P[name]) == 1, name
    T.test_decompress_programs(sim, golden)
    T.test_hash_to_g2_program(sim, oracle, golden, testdata)
    for name in ('H2C_NA', 'H2C_NM', 'H2C_NB'):          # round 6: the SWU square root by the norm method
        assert sim.nbls_sim_has_aot(vmsim_py.P[name]) == 1, name
    T.test_hash_to_g2_norm_method(sim, oracle, golden, testdata)
    T.test_norm_method_square_root_corner_cases(sim)


def test_slot_placement_table_is_current_and_pays(sim):
    """round 5 (csrc/aot_layout.h): the generated slot placements of the pairing path's programs belong to the programs as they compile NOW (a row whose hash no longer
    matches is ignored at run time -- correct, but the kernels then run with round 4's bank conflicts: regenerate with `make -C noble-bls12-381_amd/csrc layout`), and under the
    LDS bank model of the MI355X guide they remove at least a quarter of the conflict cycles of the three hot programs.  The pipelines above ran WITH these placements."""
    for name, max_frac in (('EXPX', 0.36), ('ACC_FE', 0.42), ('LINES_PQ', 0.34), ('ACC4_RAW', 0.34), ('MILLER_FE', 0.40), ('ACC2_RAW', 0.40), ('ACC_RAW', 0.42)):
        o = (C.c_ulong * 4)()
        assert sim.nbls_sim_layout_info(vmsim_py.P[name], o) == 0
        has, c0, c1, floor = list(o)
        assert has == 1, '%s: aot_layout.inc is stale for this program' % name
        assert c1 < c0 and (c1 - floor) / c1 <= max_frac and (c1 - floor) <= 0.78 * (c0 - floor), (name, c0, c1, floor)
    # the compiled placement reproduces the measured conflict fractions of round 4 (profiles/round4_pmc_b65536.json: EXPX 0.50, ACC_FE 0.52, LINES_PQ 0.44)
    for name, lo, hi in (('EXPX', 0.44, 0.54), ('ACC_FE', 0.47, 0.57), ('LINES_PQ', 0.34, 0.48)):
        o = (C.c_ulong * 4)()
        sim.nbls_sim_layout_info(vmsim_py.P[name], o)
        assert lo <= (o[1] - o[3]) / o[1] <= hi, (name, list(o))


def test_fixed_base_keys_translated(sim, oracle, testdata):
    """getPublicKey's fixed-base program (round 5) through its ahead-of-time translation: a table shared by every item (buffer stride 0) and K_SEL / K_FLAG / K_BIT steps"""
    assert sim.nbls_sim_has_aot(vmsim_py.P['G1_MUL_FIXED']) == 1
    T.test_fixed_base_public_keys(sim, oracle, testdata)


def test_gls_ladder_translated(sim, oracle):
    """sign's psi-split ladder (round 5) through its ahead-of-time translation"""
    assert sim.nbls_sim_has_aot(vmsim_py.P['G2_MUL_GLS']) == 1
    T.test_gls_ladder_for_subgroup_points(sim, oracle)


def test_sign_aligned_ladder_translated(sim, oracle):
    """sign's one-addition-per-bit ladder (round 5, pt_mul_sac_g2) through its ahead-of-time translation"""
    assert sim.nbls_sim_has_aot(vmsim_py.P['G2_MUL_SAC']) == 1
    T.test_sign_aligned_ladder_for_subgroup_points(sim, oracle)


def test_two_lane_point_chains(sim, oracle, golden, testdata):
    """round 5: the G2 point chains of a single verify / sign in their two-lane forms (nbls_aot_g2pt_ls2: clearCofactor's two ladders, sign's sign-aligned ladder; launches of at
    most 4096 items) -- the reference-generated hash vectors, the RFC 9380 suite and the structured keys of the ladder test through them"""
    for name in ('H2C_C1_LS2', 'H2C_C2_LS2', 'G2_MUL_SAC_LS2'):
        assert sim.nbls_sim_has_aot(vmsim_py.P[name]) == 1, name
    T.test_hash_to_g2_program(sim, oracle, golden, testdata, True)
    T.test_sign_aligned_ladder_for_subgroup_points(sim, oracle, 'G2_MUL_SAC_LS2')
