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
    F = C.create_string_buffer(576 * n)
    N = C.create_string_buffer(48 * n)
    out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'MILLER_FE', n, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 3: (F, 576), 4: (N, 48)})
    vmsim_py.final_exp(sim, n, F, N, out)
    for i in range(n):
        assert out.raw[576 * i:576 * (i + 1)] == hx(golden['pairs'][i]['pairing']), i


def test_final_exp_and_product(sim, oracle, golden, testdata):
    # final_exp_batch path: wire bytes -> F, N -> inverse -> hard part
    fin = hx(testdata['finalexp_in']) + hx(golden['fp12'][0]['a'])
    n = 2
    F = C.create_string_buffer(576 * n); N = C.create_string_buffer(48 * n)
    out = C.create_string_buffer(576 * n)
    vmsim_py.run(sim, 'NORM_BYTES', n, {2: (C.create_string_buffer(fin, len(fin)), 576), 3: (F, 576), 4: (N, 48)})
    vmsim_py.final_exp(sim, n, F, N, out)
    assert out.raw[:576] == hx(testdata['finalexp_out'])
    assert out.raw[576:] == hx(golden['fp12'][0]['finalexp'])
    # product of two Miller values then shared final exponentiation (golden 'product')
    p = golden['product']
    g1 = hx(''.join(p['g1'])); g2 = hx(''.join(p['g2']))
    F2 = C.create_string_buffer(576 * 2); Fp = C.create_string_buffer(576); outb = C.create_string_buffer(576)
    vmsim_py.run(sim, 'MILLER_RAW', 2, {0: (C.create_string_buffer(g1, len(g1)), 96), 1: (C.create_string_buffer(g2, len(g2)), 192), 3: (F2, 576)})
    vmsim_py.run(sim, 'MUL2', 1, {3: (F2, 1152), 5: (Fp, 576)})
    vmsim_py.run(sim, 'RAW_TO_BYTES', 1, {3: (Fp, 576), 2: (outb, 576)})
    assert outb.raw == hx(p['miller_product'])
    N1 = C.create_string_buffer(48)
    vmsim_py.run(sim, 'NORM_RAW', 1, {3: (Fp, 576), 4: (N1, 48)})
    vmsim_py.final_exp(sim, 1, Fp, N1, outb)
    assert outb.raw == hx(p['result'])
