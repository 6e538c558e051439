"""Debug / sanitizer build (`make -C noble-bls12-381_amd/csrc debug`).

CPU: the static program verifier accepts every compiled step program and catches damaged ones; the host compiler + simulator run clean under
AddressSanitizer / UBSan (nbls_selftest) and reproduce the reference's Miller value of the generators.
GPU: the checked engine library (libnbls_dbg.so: verification before upload, buffer-extent checks at every launch) passes the golden pairings."""
import ctypes as C
import importlib
import os
import subprocess
import sys
import pytest
import vmsim_py
from goldenio import hx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'noble-bls12-381_amd', 'csrc')
# pairing(G1.BASE, G2.BASE, false).c0.c0.c0 as produced by the reference
MILLER_ANCHOR = '0d884dea7038a532183bc322e31bd35b79cddaccb161bc7ede312491c647b146131f1e4bff594b072bd566bb02d87fe3'


def test_verifier_accepts_every_program():
    sim = vmsim_py.load()
    msg = C.create_string_buffer(256)
    n = sim.nbls_sim_program_count()
    assert n >= 60
    for k in range(n):
        assert sim.nbls_sim_verify(k, msg, 256) == 0, msg.value.decode()


def test_verifier_catches_damage():
    sim = vmsim_py.load()
    msg = C.create_string_buffer(256)
    P = vmsim_py.P
    cases = [
        (P['MILLER_FE'], 8 * 16 * 3 + 8, 0x00010000, -1, 'term'),          # a product term pushed 64 KB out of the instance region
        (P['MILLER_FE'], 8 * 16 * 3 + 8, 0x00000004, -1, 'term'),          # ... or off its 16-byte alignment
        (P['EXPX'], 2, 0x7fffff00, 0, 'descriptors outside'),              # step 2 reads its descriptors beyond the program
        (P['EXPX'], 5, 0x40, 1, 'active lanes'),                            # more active lanes than the instance has
        (P['G1_VALIDATE'], 3, 0x40, 3, 'step kind'),                        # an unknown step kind
    ]
    for prog, word, flip, field, expect in cases:
        r = sim.nbls_sim_verify_damaged(prog, word, flip, field, msg, 256)
        assert r == 1, (prog, word, hex(flip), field)
        assert expect in msg.value.decode() or msg.value, msg.value
    # destination inside the constant region: word 0 of the first DOT lane of a program = dst offset
    r = sim.nbls_sim_verify_damaged(P['FP12_MUL2'] if 'FP12_MUL2' in P else P['MUL2'], 0, 0, -1, msg, 256)
    assert r == 0      # undamaged copy is clean


@pytest.mark.skipif(subprocess.call(['sh', '-c', 'echo "int main(){}" | g++ -x c++ -fsanitize=address,undefined -o /dev/null - 2>/dev/null']) != 0, reason='no sanitizer runtime')
def test_sanitizer_selftest():
    subprocess.check_call(['make', '-s', '-C', CSRC, '../nbls_selftest'])
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0')
    out = subprocess.run([os.path.join(ROOT, 'noble-bls12-381_amd', 'nbls_selftest')], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'failures 0' in out.stdout and ('miller c0.c0.c0 ' + MILLER_ANCHOR) in out.stdout, out.stdout
    assert 'ERROR: AddressSanitizer' not in out.stderr and 'runtime error' not in out.stderr, out.stderr


@pytest.mark.gpu
def test_checked_engine_library(golden):
    """the golden pairings, a Miller product and a verifyBatch through the checked library in a fresh process (NBLS_LIBRARY selects it)"""
    lib = os.path.join(ROOT, 'noble-bls12-381_amd', 'libnbls_dbg.so')
    if not os.path.exists(lib):
        subprocess.check_call(['make', '-s', '-C', CSRC, '../libnbls_dbg.so'])
    code = r'''
import sys, importlib, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, goldenio, oracle_py
from goldenio import hx
pkg = importlib.import_module('noble-bls12-381_amd')
g = goldenio.load('ref_vectors.json.gz'); o = oracle_py.load(rebuild=False)
eng = pkg.Engine(0)
g1 = b''.join(hx(v['g1']) for v in g['pairs']); g2 = b''.join(hx(v['g2']) for v in g['pairs'])
for split in (1 << 40, 0):
    eng.set_split_miller_min(split)
    out, _ = eng.pairing_batch(g1, g2, True, True)
    assert all(out[576 * i:576 * (i + 1)] == hx(v['pairing']) for i, v in enumerate(g['pairs']))
    assert eng.miller_product(g1, g2, True)[0] == o.miller_product(g1, g2, final_exp=True)
vb = g['verify_batch']
assert eng.verify_batch(hx(vb['agg_sig']), [hx(m) for m in vb['msgs']], [hx(k) for k in vb['pks']]) is True
# a launch whose buffer is smaller than what the program touches is refused by the checked library
r = eng.lib.nbls_pairing_prepared_dev(eng.h, 1, C.c_void_p(1), C.c_void_p(1), 64, 0, C.c_void_p(1), None)
assert r == -1, r
print('checked ok')
''' % (ROOT, os.path.join(ROOT, 'tests'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=dict(os.environ, NBLS_LIBRARY=lib))
    assert out.returncode == 0 and 'checked ok' in out.stdout, out.stdout + out.stderr
