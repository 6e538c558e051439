"""GPU parity tests: the HIP engine through the C ABI (libnbls.so) vs the CPU oracle and the golden vectors."""
import hashlib
import importlib
import os
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from goldenio import hx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    pkg = importlib.import_module('noble-bls12-381_amd')
    return pkg.Engine(0)


def _rand_points(oracle, n, seed):
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    G1, G2 = [], []
    for i in range(n):
        a = int.from_bytes(hashlib.sha256(b'nbls-test-%d-%d-a' % (seed, i)).digest(), 'big') % (2**250) + 1
        b = int.from_bytes(hashlib.sha256(b'nbls-test-%d-%d-b' % (seed, i)).digest(), 'big') % (2**250) + 1
        G1.append(oracle.g1_mul(g1, a)[1])
        G2.append(oracle.g2_mul(g2, b)[1])
    return b''.join(G1), b''.join(G2)


def test_golden_pairs(eng, golden):
    g1 = b''.join(hx(v['g1']) for v in golden['pairs'])
    g2 = b''.join(hx(v['g2']) for v in golden['pairs'])
    n = len(golden['pairs'])
    out, st = eng.pairing_batch(g1, g2, True, False)
    ml, _ = eng.pairing_batch(g1, g2, False, False)
    for i, v in enumerate(golden['pairs']):
        assert out[576 * i:576 * (i + 1)] == hx(v['pairing']), i
        assert ml[576 * i:576 * (i + 1)] == hx(v['miller']), i


def test_reference_kats(eng, oracle, testdata):
    out, _ = eng.pairing_batch(oracle.g1_generator(), oracle.g2_generator(), True, False)
    assert out == hx(testdata['e_G1_G2'])                                   # test/pairing.test.ts:46-64
    assert eng.final_exp_batch(hx(testdata['finalexp_in'])) == hx(testdata['finalexp_out'])   # test/pairing.test.ts:65-96


def test_kilic_vectors(eng, oracle, testdata):
    """test/deterministic.test.ts:34-46: e(i*G1, i*G2), first 128 of the 1000 kilic vectors."""
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    n = 128
    G1 = b''.join(oracle.g1_mul(g1, i)[1] for i in range(1, n + 1))
    G2 = b''.join(oracle.g2_mul(g2, i)[1] for i in range(1, n + 1))
    out, _ = eng.pairing_batch(G1, G2, True, False)
    for i in range(n):
        assert out[576 * i:576 * (i + 1)] == hx(testdata['pairing_iG1_iG2'][i]), i


@pytest.mark.parametrize('n', [1, 2, 3, 63, 64, 65, 200])
def test_random_batches_vs_oracle(eng, oracle, n):
    G1, G2 = _rand_points(oracle, n, n)
    out, _ = eng.pairing_batch(G1, G2, True, False)
    ref, _ = oracle.pairing_batch(G1, G2, True, False, threads=16)
    assert out == ref
    out, _ = eng.pairing_batch(G1, G2, False, False)
    ref, _ = oracle.pairing_batch(G1, G2, False, False, threads=16)
    assert out == ref


def test_empty_batch(eng):
    out, st = eng.pairing_batch(b'', b'', True, False)
    assert out == b'' and st == b''


@pytest.mark.parametrize('n', [1, 2, 3, 5, 8, 33])
def test_miller_product(eng, oracle, golden, n):
    G1, G2 = _rand_points(oracle, n, 1000 + n)
    for fe in (False, True):
        out, _ = eng.miller_product(G1, G2, fe, False)
        assert out == oracle.miller_product(G1, G2, fe)


def test_product_golden(eng, golden):
    p = golden['product']
    out, _ = eng.miller_product(hx(''.join(p['g1'])), hx(''.join(p['g2'])), True, False)
    assert out == hx(p['result'])
    out, _ = eng.miller_product(hx(''.join(p['g1'])), hx(''.join(p['g2'])), False, False)
    assert out == hx(p['miller_product'])


def test_final_exp_batch(eng, oracle, golden):
    ins = b''.join(hx(v['a']) for v in golden['fp12'])
    out = eng.final_exp_batch(ins)
    for i, v in enumerate(golden['fp12']):
        assert out[576 * i:576 * (i + 1)] == hx(v['finalexp'])


def test_batch_4096_properties(eng, oracle):
    """BASELINE config 2 size: 4096 pairings.  Full comparison against the multi-threaded oracle, plus bilinearity as a
    size-independent property: prod_i e(P_i, Q_i) * e(-P_i, Q_i) == 1 via the shared-final-exponentiation path."""
    n = 4096
    base1, base2 = _rand_points(oracle, 64, 77)
    G1 = base1 * (n // 64)
    G2 = b''.join(base2[192 * ((i * 7 + i // 64) % 64):192 * ((i * 7 + i // 64) % 64 + 1)] for i in range(n))
    out, _ = eng.pairing_batch(G1, G2, True, False)
    ref, _ = oracle.pairing_batch(G1, G2, True, False, threads=min(64, os.cpu_count() or 8))
    assert hashlib.sha256(out).hexdigest() == hashlib.sha256(ref).hexdigest()
    assert out == ref


@pytest.mark.parametrize('n', [1023, 1024, 1025, 1029, 4095, 4097, 16383, 16384, 16385])
def test_kernel_selection_boundaries(eng, oracle, n):
    """batch sizes around the launch-shape decisions of the runtime: 256 workgroups (two-wave kernel vs one-wave kernel),
    partially filled last waves (4 items per wave in the Miller programs, 5 in EXPX), odd / even pair counts of the shared
    accumulator Miller program -- full comparison with the multi-threaded oracle"""
    base1, base2 = _rand_points(oracle, 32, 4242)
    G1 = b''.join(base1[96 * ((i * 5 + 1) % 32):96 * ((i * 5 + 1) % 32) + 96] for i in range(n))
    G2 = b''.join(base2[192 * ((i // 32 + i * 3) % 32):192 * ((i // 32 + i * 3) % 32) + 192] for i in range(n))
    out, _ = eng.pairing_batch(G1, G2, True, False)
    ref, _ = oracle.pairing_batch(G1, G2, True, False, threads=min(64, os.cpu_count() or 8))
    assert out == ref
    if n <= 4097:      # the single-threaded oracle product is the slow part
        assert eng.miller_product(G1, G2, True)[0] == oracle.miller_product(G1, G2, True)


def test_batch_131072_properties(eng, oracle):
    """BASELINE config 4 size per GPU (1M pairings over 8 GPUs): 131072 pairings in one call.  The oracle cannot follow at this
    size, so: (i) the inputs cycle through 64 x 64 distinct (P_a, Q_b) combinations, every repetition of a combination must give
    the identical 576 bytes wherever it sits in the batch; (ii) the 4096 distinct results are compared with the oracle;
    (iii) e(P_a, Q_b) * e(-P_a, Q_b) == 1 over the whole batch through the shared-final-exponentiation path."""
    n = 131072
    base1, base2 = _rand_points(oracle, 64, 131)
    idx = [((i * 13 + 5) % 64, (i // 64 + i * 7) % 64) for i in range(n)]
    G1 = b''.join(base1[96 * a:96 * a + 96] for a, _ in idx)
    G2 = b''.join(base2[192 * b:192 * b + 192] for _, b in idx)
    out, st = eng.pairing_batch(G1, G2, True, False)
    assert st == bytes(n)
    first = {}
    for i, key in enumerate(idx):
        o = out[576 * i:576 * i + 576]
        if key in first:
            assert o == first[key], i
        else:
            first[key] = o
    keys = sorted(first)
    ref, _ = oracle.pairing_batch(b''.join(base1[96 * a:96 * a + 96] for a, _ in keys), b''.join(base2[192 * b:192 * b + 192] for _, b in keys), True, False,
                                  threads=min(64, os.cpu_count() or 8))
    assert b''.join(first[k] for k in keys) == ref
    neg = b''.join(oracle.un('g1_neg_aff', base1[96 * a:96 * a + 96], 96) for a in range(64))
    G1n = b''.join(neg[96 * a:96 * a + 96] for a, _ in idx)
    prod, _ = eng.miller_product(G1 + G1n, G2 + G2, True)
    assert prod == bytes(47) + b'\x01' + bytes(528)


@pytest.mark.parametrize('ls_max', ['0', '1024'])
def test_plain_and_lane_split_programs(ls_max):
    """Launches of at most 1024 items run the lane-split programs (ahead-of-time kernels nbls_aot_miller_ls / nbls_aot_expx_ls, interpreter nbls_vm_kernel_ls4: every K_DOT lane-op on four adjacent lanes, columns summed by DPP); NBLS_LS_MAX=0
    keeps the plain programs for them.  Both in turn over the whole pipeline: pairings at several batch sizes against the oracle, a Miller product and a verifyBatch.
    NBLS_LS_MAX is read once per process, hence the subprocess."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import importlib, os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
        import oracle_py, goldenio
        from goldenio import hx
        pkg = importlib.import_module('noble-bls12-381_amd')
        eng, oracle, golden = pkg.Engine(0), oracle_py.load(), goldenio.load('ref_vectors.json.gz')
        pairs = golden['pairs']
        for n in (1, 3, 37, 1024, 1500):
            g1 = b''.join(hx(pairs[i %% len(pairs)]['g1']) for i in range(n)); g2 = b''.join(hx(pairs[(5 * i + 1) %% len(pairs)]['g2']) for i in range(n))
            for fe in (True, False):
                out, st = eng.pairing_batch(g1, g2, fe, False)
                ref, _ = oracle.pairing_batch(g1, g2, fe, False, threads=16)
                assert out == ref, n
            assert eng.miller_product(g1, g2, True)[0] == oracle.miller_product(g1, g2, True), n
        vb = golden['verify_batch']
        assert eng.verify_batch(hx(vb['agg_sig']), [hx(m) for m in vb['msgs']], [hx(p) for p in vb['pks']]) is True
        names = [eng.lib.nbls_program_name(i).decode() for i in range(eng.lib.nbls_program_count())]
        assert 'expx_ls' in names
        print('ok')
    ''') % (ROOT, ROOT)
    env = dict(os.environ, NBLS_LS_MAX=ls_max)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


def test_batches_in_flight(oracle):
    """PairingPipeline: several engine contexts, one stream each, batches submitted back to back without waiting -- every
    batch's results must equal the oracle's although their kernels overlap on the GPU"""
    import torch
    pkg = importlib.import_module('noble-bls12-381_amd')
    pipe = pkg.PairingPipeline(0, 3)
    n, rounds = 700, 7
    sets = [_rand_points(oracle, 16, 9000 + r) for r in range(rounds)]
    ins, outs = [], []
    for r in range(rounds):
        G1 = b''.join(sets[r][0][96 * (i % 16):96 * (i % 16) + 96] for i in range(n))
        G2 = b''.join(sets[r][1][192 * ((3 * i + 1) % 16):192 * ((3 * i + 1) % 16) + 192] for i in range(n))
        ins.append((G1, G2, torch.frombuffer(bytearray(G1), dtype=torch.uint8).cuda(), torch.frombuffer(bytearray(G2), dtype=torch.uint8).cuda()))
        outs.append(torch.empty(576 * n, dtype=torch.uint8, device='cuda'))
    torch.cuda.synchronize()
    for r in range(rounds):
        pipe.submit(n, ins[r][2].data_ptr(), ins[r][3].data_ptr(), outs[r].data_ptr(), True)
    pipe.synchronize()
    for r in range(rounds):
        ref, _ = oracle.pairing_batch(ins[r][0], ins[r][1], True, False, threads=16)
        assert bytes(outs[r].cpu().numpy().tobytes()) == ref, r


def test_two_halves_execution(oracle, golden):
    """nbls_pairing_batch_dev runs batches from 16,384 pairs as two halves on two streams sharing the caller's scratch through an item offset; forced here at a
    size the oracle can check: same bytes as the one-stream execution, with and without the final exponentiation, odd and even sizes."""
    pkg = importlib.import_module('noble-bls12-381_amd')
    eng = pkg.Engine(0)
    for n in (130, 517):
        g1 = b''.join(hx(golden['pairs'][i % len(golden['pairs'])]['g1']) for i in range(n)); g2 = b''.join(hx(golden['pairs'][(7 * i + 3) % len(golden['pairs'])]['g2']) for i in range(n))
        for fe in (True, False):
            eng.set_halves_min(0)
            ref, _ = eng.pairing_batch(g1, g2, fe, False)
            eng.set_halves_min(64)
            got, _ = eng.pairing_batch(g1, g2, fe, False)
            assert got == ref
            exp, _ = oracle.pairing_batch(g1, g2, fe, False, threads=16)
            assert got == exp
    eng.set_halves_min(16384)


def test_chained_and_separate_final_exponentiation(oracle, golden):
    """The middle of the final exponentiation (EXPX, FE_MID1, EXPX x 3, FE_MID2, EXPX; math.ts:862-867) is one chained launch below NBLS_TUNE_CHAIN_MAX items and seven launches
    otherwise (what PairingPipeline's contexts use): same bytes either way, equal to the oracle's; sizes above the lane-split ranges (1024 / 2048: those forms run unchained), with a partly filled last wavefront."""
    pkg = importlib.import_module('noble-bls12-381_amd')
    eng = pkg.Engine(0)
    pairs = golden['pairs']
    for n in (2051, 3077):
        g1 = b''.join(hx(pairs[(5 * i + 2) % len(pairs)]['g1']) for i in range(n)); g2 = b''.join(hx(pairs[(3 * i + n) % len(pairs)]['g2']) for i in range(n))
        eng.set_chain_max(8192)
        eng.timing_enable(True)
        chained, _ = eng.pairing_batch(g1, g2, True, False)
        launches_chained = sum(int(v[1]) for v in eng.timing_read().values())
        eng.set_chain_max(0)
        separate, _ = eng.pairing_batch(g1, g2, True, False)
        launches_separate = sum(int(v[1]) for v in eng.timing_read().values())     # reading clears the counts
        eng.timing_enable(False)
        assert chained == separate
        checked = bool(os.environ.get('NBLS_CHECKED')) or 'dbg' in os.environ.get('NBLS_LIBRARY', '')     # checked mode keeps one launch per program (its buffer checks live in run())
        assert launches_separate == launches_chained + (0 if checked else 6), (launches_chained, launches_separate)     # seven launches instead of one
        exp, _ = oracle.pairing_batch(g1, g2, True, False, threads=16)
        assert chained == exp
    eng.set_chain_max(8192)


@pytest.mark.gpu
def test_default_dispatch_thresholds(oracle, golden):
    """Batch sizes either side of the library's default dispatch thresholds (two halves from 8192 pairs, except the one-full-round window 10,753 .. 12,288),
    untouched tuning: every pairing byte-equal to the oracle, last item included."""
    import os
    pkg = importlib.import_module('noble-bls12-381_amd')
    eng = pkg.Engine(0)
    pairs = golden['pairs']
    for n in (8191, 8192, 10753, 12289):
        g1 = b''.join(hx(pairs[(3 * i + n) % len(pairs)]['g1']) for i in range(n)); g2 = b''.join(hx(pairs[(5 * i + 1) % len(pairs)]['g2']) for i in range(n))
        got, _ = eng.pairing_batch(g1, g2, True, False)
        exp, _ = oracle.pairing_batch(g1, g2, True, False, threads=os.cpu_count() or 16)
        assert got == exp, n


@pytest.mark.parametrize('n', [1025, 1530, 2048])
def test_two_lane_split_range(oracle, golden, n):
    """round 5: calls of 1025 .. 2048 pairs run the two-lane forms (nbls_aot_miller_ls2 / nbls_aot_expx_ls2: two items per wavefront, every K_DOT lane-op on two adjacent lanes,
    one DPP stage) -- pairings, raw Miller values and a Miller product against the oracle at both ends of the range and inside it, and the same bytes as the plain forms"""
    pkg = importlib.import_module('noble-bls12-381_amd')
    eng = pkg.Engine(0)
    b = eng.kernel_bindings()
    assert b['miller_fe_ls2'] == 'nbls_aot_miller_ls2' and b['expx_ls2'] == 'nbls_aot_expx_ls2' and b['miller_raw_ls2'] == 'nbls_aot_miller_ls2'
    pairs = golden['pairs']
    g1 = b''.join(hx(pairs[(7 * i + 1) % len(pairs)]['g1']) for i in range(n)); g2 = b''.join(hx(pairs[(3 * i + 2 * n) % len(pairs)]['g2']) for i in range(n))
    eng.timing_enable(True)
    out, _ = eng.pairing_batch(g1, g2, True, False)
    tm = eng.timing_read(); eng.timing_enable(False)
    assert 'miller_fe_ls2' in tm and 'expx_ls2' in tm and 'miller_fe' not in tm and 'expx' not in tm, sorted(tm)      # the two-lane forms are what ran
    ref, _ = oracle.pairing_batch(g1, g2, True, False, threads=32)
    assert out == ref
    ml, _ = eng.pairing_batch(g1, g2, False, False)
    refm, _ = oracle.pairing_batch(g1, g2, False, False, threads=32)
    assert ml == refm
    assert eng.miller_product(g1, g2, True)[0] == oracle.miller_product(g1, g2, True)


def test_one_limb_per_lane_forms(oracle, golden):
    """round 6: the one-limb-per-lane interpreter (nbls_vm_kernel_wide: one item per workgroup of three wavefronts, a lane-op per row of sixteen lanes, two barriers per step;
    an experiment, off by default because it measured slower than the lane-split forms) switched on for launches of at most 128 items -- final exponentiation, Miller products
    and verifyBatch on both sides of the switch against the oracle, and the same bytes with the form switched off"""
    import hashlib
    pkg = importlib.import_module('noble-bls12-381_amd')
    eng = pkg.Engine(0)
    pairs = golden['pairs']
    for n in (1, 2, 5, 127, 128, 129):
        g1 = b''.join(hx(pairs[(5 * i + 3) % len(pairs)]['g1']) for i in range(n)); g2 = b''.join(hx(pairs[(7 * i + n) % len(pairs)]['g2']) for i in range(n))
        eng.set_wide_max(128)
        out, _ = eng.pairing_batch(g1, g2, True, False)
        prod = eng.miller_product(g1, g2, True)[0]
        assert out == oracle.pairing_batch(g1, g2, True, False, threads=32)[0], n
        assert prod == oracle.miller_product(g1, g2, True), n
        eng.set_wide_max(0)
        assert eng.pairing_batch(g1, g2, True, False)[0] == out and eng.miller_product(g1, g2, True)[0] == prod, n
    eng.set_wide_max(128)
    # raw Fp12 inputs through the final exponentiation alone: random elements, ONE, and the all-(p - 1) element
    rnd = __import__("random").Random(128)
    P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    elems = [[rnd.randrange(P) for _ in range(12)] for _ in range(6)] + [[1] + [0] * 11, [P - 1] * 12]
    blob = b''.join(b''.join(v.to_bytes(48, 'big') for v in e) for e in elems)
    got = eng.final_exp_batch(blob)
    for i in range(len(elems)):
        assert got[576 * i:576 * i + 576] == oracle.un('fp12_final_exp', blob[576 * i:576 * i + 576], 576), i
    # one verify and a small verifyBatch (their tails run the one-element final exponentiation)
    sks = [(int.from_bytes(hashlib.sha256(b'wide-sk%d' % i).digest(), 'big') % (1 << 254) + 1).to_bytes(32, 'big') for i in range(9)]
    msgs = [hashlib.sha256(b'wide-m%d' % i).digest() for i in range(9)]
    pk, sig = oracle.aggregate_sign(msgs, sks, threads=8)
    assert eng.verify_batch(sig, msgs, pk) is True and eng.verify_batch(sig, msgs[::-1], pk) is False
    assert eng.verify_batch(oracle.sign(msgs[0], sks[0])[1], msgs[:1], pk[:1]) is True
