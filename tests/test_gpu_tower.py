"""Known-answer tests of the single tower operations on the device (nbls_tower_op_batch, include/nbls.h): the vectors tools/gen_golden.mjs produced by running the
reference itself (Fp / Fp2 / Fp6 / Fp12 add, subtract, multiply, square, invert, Frobenius maps, conjugate, multiplication by the non-residue, the sparse products
multiplyBy1 / 01 / 014, cyclotomicSquare, cyclotomicExp: math.ts:223-273, 451-539, 601-688, 732-852), plus algebraic identities where the file holds no vector
(Frobenius powers composed from the first one, a * a^-1 = 1, unitary inverse = conjugate).  Square roots are covered through the decoders (tests/test_gpu_codec.py)."""
import importlib
import pytest
from goldenio import hx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine():
    pkg = importlib.import_module('noble-bls12-381_amd')
    return pkg.Engine(0)

ADD, SUB, NEG, MUL, SQR, INV, FROB, CONJ, MULNR, MULB, MUL1, MUL01, MUL014, CYCSQR, CYCEXP = range(15)
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def cat(vs, key):
    return b''.join(hx(v[key]) for v in vs)


def test_fp_ops(engine, golden):
    vs = [v for v in golden['fp'] if v['inv'] is not None]
    a, b = cat(vs, 'a'), cat(vs, 'b')
    for op, key in ((ADD, 'add'), (SUB, 'sub'), (MUL, 'mul')):
        assert engine.tower_op(1, op, a, b) == cat(vs, key), key
    for op, key in ((NEG, 'neg'), (SQR, 'sqr'), (INV, 'inv')):
        assert engine.tower_op(1, op, a) == cat(vs, key), key
    # the zero element and p - 1: negate / square / multiply at the edges of the range
    edge = (0).to_bytes(48, 'big') + (P - 1).to_bytes(48, 'big') + (1).to_bytes(48, 'big')
    assert engine.tower_op(1, NEG, edge) == (0).to_bytes(48, 'big') + (1).to_bytes(48, 'big') + (P - 1).to_bytes(48, 'big')
    assert engine.tower_op(1, SQR, edge) == (0).to_bytes(48, 'big') + (1).to_bytes(48, 'big') + (1).to_bytes(48, 'big')
    assert engine.tower_op(1, INV, edge[48:]) == (P - 1).to_bytes(48, 'big') + (1).to_bytes(48, 'big')


def test_fp2_ops(engine, golden):
    vs = golden['fp2']
    a, b = cat(vs, 'a'), cat(vs, 'b')
    for op, key in ((ADD, 'add'), (SUB, 'sub'), (MUL, 'mul')):
        assert engine.tower_op(2, op, a, b) == cat(vs, key), key
    for op, key in ((SQR, 'sqr'), (INV, 'inv'), (MULNR, 'mulnr'), (MULB, 'mulB')):
        assert engine.tower_op(2, op, a) == cat(vs, key), key
    assert engine.tower_op(2, FROB, a, param=1) == cat(vs, 'frob1')
    assert engine.tower_op(2, CONJ, a) == cat(vs, 'frob1')
    assert engine.tower_op(2, FROB, a, param=2) == a


def test_fp6_ops(engine, golden):
    vs = golden['fp6']
    a, b = cat(vs, 'a'), cat(vs, 'b')
    assert engine.tower_op(6, MUL, a, b) == cat(vs, 'mul')
    for op, key in ((SQR, 'sqr'), (INV, 'inv'), (MULNR, 'mulnr')):
        assert engine.tower_op(6, op, a) == cat(vs, key), key
    assert engine.tower_op(6, MUL1, a, cat(vs, 'b1')) == cat(vs, 'mul1')
    assert engine.tower_op(6, MUL01, a, cat(vs, 'b0'), cat(vs, 'b1')) == cat(vs, 'mul01')
    for k in range(1, 6):
        assert engine.tower_op(6, FROB, a, param=k) == b''.join(hx(v['frob'][k - 1]) for v in vs), k
    assert engine.tower_op(6, FROB, a, param=6) == a
    # a * a^-1 = 1
    one = (1).to_bytes(48, 'big') + bytes(5 * 48)
    assert engine.tower_op(6, MUL, a, cat(vs, 'inv')) == one * len(vs)


def test_fp12_ops(engine, golden):
    vs = golden['fp12']
    a, b = cat(vs, 'a'), cat(vs, 'b')
    assert engine.tower_op(12, MUL, a, b) == cat(vs, 'mul')
    for op, key in ((SQR, 'sqr'), (INV, 'inv'), (CONJ, 'conj')):
        assert engine.tower_op(12, op, a) == cat(vs, key), key
    assert engine.tower_op(12, MUL014, a, cat(vs, 'o0'), cat(vs, 'o1'), cat(vs, 'o4')) == cat(vs, 'mul014')
    for i, k in enumerate((1, 2, 3, 6)):
        assert engine.tower_op(12, FROB, a, param=k) == b''.join(hx(v['frob'][i]) for v in vs), k
    # Frobenius maps the golden file does not hold: phi^k composed from phi^1, and phi^12 = identity
    x = a
    for k in range(1, 13):
        x = engine.tower_op(12, FROB, x, param=1)
        if k < 12:
            assert engine.tower_op(12, FROB, a, param=k) == x, k
    assert x == a
    u = cat(vs, 'unitary')
    assert engine.tower_op(12, CYCSQR, u) == cat(vs, 'cyclosqr')
    assert engine.tower_op(12, CYCEXP, u) == cat(vs, 'cycloexp')
    # on unitary elements the cyclotomic square is the square and the inverse is the conjugate
    assert engine.tower_op(12, SQR, u) == cat(vs, 'cyclosqr')
    assert engine.tower_op(12, INV, u) == engine.tower_op(12, CONJ, u)
    # a non-unitary inverse: a * a^-1 = 1
    one = (1).to_bytes(48, 'big') + bytes(11 * 48)
    assert engine.tower_op(12, MUL, a, cat(vs, 'inv')) == one * len(vs)


def test_unsupported_combinations_are_rejected(engine, golden):
    import importlib
    pkg = importlib.import_module('noble-bls12-381_amd')
    a = cat(golden['fp6'], 'a')
    for field, op in ((6, CONJ), (1, FROB), (2, MUL014), (12, MULB), (7, ADD)):
        with pytest.raises(pkg.NblsError):
            engine.tower_op(field, op, a[:48 * field] if field != 7 else a[:48])
