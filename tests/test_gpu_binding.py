"""Round 5: which kernels run the step programs on THIS box, and parity of the shipped fallback.

(i) every program of the pairing / verifyBatch / sign / MSM paths is bound to its ahead-of-time kernel (`nbls_program_kernel`): a host-compiler mismatch on a future box
    would otherwise benchmark the interpreter (-25 %) with green tests; (ii) the interpreter (`NBLS_AOT=0`: what runs on such a mismatch) is re-run in a subprocess on the golden
    pairs (reference-generated Miller values and pairings), a 1,500-pair batch against the oracle and the reference's verifyBatch fixture; (iii) round 4's two-phase verifyBatch
    (`NBLS_VERIFY_PIPE=0`, kept for A/B runs) on the same fixture.  SURVEY 8(c)."""
import importlib
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

# programs that have no ahead-of-time kernel by design: the compressed-squaring experiment (off by default), test-only pieces of hash-to-G2, round 3's H2C_B / H2C_C
INTERPRETER_ONLY = {'expc_sq', 'expc_dec_a', 'expc_dec_b', 't_swu', 't_iso', 't_clear', 'h2c_b', 'h2c_c'}


def test_hot_programs_run_on_ahead_of_time_kernels():
    pkg = importlib.import_module('noble-bls12-381_amd')
    eng = pkg.Engine(0)
    b = eng.kernel_bindings()
    on_vm = sorted(p for p, k in b.items() if not k.startswith('nbls_aot_'))
    assert set(on_vm) <= INTERPRETER_ONLY, 'step programs on the interpreter that should have an ahead-of-time kernel: %s' % sorted(set(on_vm) - INTERPRETER_ONLY)
    # the product paths by name
    for p, k in (('lines_pq', 'nbls_aot_lines_pq'), ('acc_fe', 'nbls_aot_acc_fe'), ('acc4_raw', 'nbls_aot_acc4_raw'), ('expx', 'nbls_aot_expx'), ('fe_easy', 'nbls_aot_fe_easy'),
                 ('fe_final', 'nbls_aot_fe_final'), ('miller_fe', 'nbls_aot_miller_fe'), ('miller_fe_ls', 'nbls_aot_miller_ls'), ('expx_ls', 'nbls_aot_expx_ls'),
                 ('fp12_mul2s', 'nbls_aot_mul2'), ('h2c_c1', 'nbls_aot_h2c_c1'), ('g1_dec_b', 'nbls_aot_g1_dec')):
        assert b[p] == k, (p, b[p])
    assert len(b) - len(on_vm) >= 76


CHILD = r'''
import hashlib, importlib, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch
import goldenio, oracle_py
from goldenio import hx
pkg = importlib.import_module('noble-bls12-381_amd')
eng = pkg.Engine(0)
mode = sys.argv[1]
b = eng.kernel_bindings()
if mode == 'interp':
    # every program on the interpreter; the two-lane programs have no interpreter form and say so (round 6: they used to be reported as 'nbls_vm_kernel_ls4')
    assert all(k.startswith('nbls_vm_kernel') or (p.endswith('_ls2') and k.startswith('none')) for p, k in b.items()), [p for p, k in b.items() if not k.startswith('nbls_vm_kernel')]
    assert all(k.startswith('none') for p, k in b.items() if p.endswith('_ls2'))
golden = goldenio.load('ref_vectors.json.gz')
oracle = oracle_py.load()
g1 = b''.join(hx(v['g1']) for v in golden['pairs']); g2 = b''.join(hx(v['g2']) for v in golden['pairs'])
out, _ = eng.pairing_batch(g1, g2, True, False); ml, _ = eng.pairing_batch(g1, g2, False, False)
for i, v in enumerate(golden['pairs']):
    assert out[576 * i:576 * (i + 1)] == hx(v['pairing']), i
    assert ml[576 * i:576 * (i + 1)] == hx(v['miller']), i
n = 1500
G1, G2 = [], []
for i in range(64):
    a = int.from_bytes(hashlib.sha256(b'fallback-a%%d' %% i).digest(), 'big') %% (2 ** 250) + 1
    c = int.from_bytes(hashlib.sha256(b'fallback-b%%d' %% i).digest(), 'big') %% (2 ** 250) + 1
    G1.append(oracle.g1_mul(oracle.g1_generator(), a)[1]); G2.append(oracle.g2_mul(oracle.g2_generator(), c)[1])
P = b''.join(G1[i %% 64] for i in range(n)); Q = b''.join(G2[(5 * i + i // 64) %% 64] for i in range(n))
for fe in (True, False):
    got, _ = eng.pairing_batch(P, Q, fe, False)
    ref, _ = oracle.pairing_batch(P, Q, fe, False, threads=32)
    assert got == ref, fe
eng.set_split_miller_min(0)      # the two-program Miller loop as well
got, _ = eng.pairing_batch(P, Q, True, False)
assert got == oracle.pairing_batch(P, Q, True, False, threads=32)[0]
vb = golden['verify_batch']
msgs, pks = [hx(m) for m in vb['msgs']], [hx(p) for p in vb['pks']]
assert eng.verify_batch(hx(vb['agg_sig']), msgs, pks) is True
m2 = list(msgs); m2[1] = m2[1][:3] + bytes([m2[1][3] ^ 1]) + m2[1][4:]
assert eng.verify_batch(hx(vb['agg_sig']), m2, pks) is False
# a batch large enough for the chunked pipeline / the two-program product: 96 signers, keys and signatures from the oracle
sks = [(int.from_bytes(hashlib.sha256(b'fb-sk%%d' %% i).digest(), 'big') %% (2 ** 254) + 1).to_bytes(32, 'big') for i in range(96)]
ms = [hashlib.sha256(b'fb-msg%%d' %% i).digest() for i in range(96)]
pk, sig = oracle.aggregate_sign(ms, sks, threads=32)
eng.set_verify_pipeline(chunks=3, last_pct=20, pipe_min=0)
assert eng.verify_batch(sig, ms, pk) is True
assert eng.verify_batch(sig, ms[::-1], pk) is False
print('CHILD_OK', mode, eng.config_describe())
'''


@pytest.mark.parametrize('mode,env', [('interp', {'NBLS_AOT': '0'}), ('twophase', {'NBLS_VERIFY_PIPE': '0'})])
def test_fallback_paths_in_a_subprocess(mode, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT}, mode], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'CHILD_OK ' + mode in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    for k, v in env.items():
        assert '%s=%s(env)' % (k, v) in r.stdout, r.stdout[-1500:]
