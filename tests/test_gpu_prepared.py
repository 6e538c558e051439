"""GPU parity of the prepared-Q path: PointG2.pairingPrecomputes (index.ts:703-711) = calcPairingPrecomputes (math.ts:1331-1371) as a value,
and PointG1.millerLoop (index.ts:452-454) / pairing over prepared tables, against the reference-generated fixtures and the oracle."""
import hashlib
import importlib
import pytest
from goldenio import hx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    pkg = importlib.import_module('noble-bls12-381_amd')
    return pkg.Engine(0)


def test_line_tables_match_reference(eng, golden):
    """the 19,584-byte table of every golden pair: first / last triple and SHA-256 as produced by the reference"""
    vs = golden['pairs']
    g2 = b''.join(hx(v['g2']) for v in vs)
    t = eng.g2_prepare(g2)
    W = eng.LINE_WIRE_BYTES
    assert len(t) == W * len(vs)
    for i, v in enumerate(vs):
        tab = t[W * i:W * (i + 1)]
        assert tab[:288] == hx(v['ell_first']) and tab[-288:] == hx(v['ell_last']), i
        assert hashlib.sha256(tab).hexdigest() == v['ell_sha256'], i


def test_prepared_pairings(eng, oracle, golden):
    vs = golden['pairs']
    n = len(vs)
    g1 = b''.join(hx(v['g1']) for v in vs); g2 = b''.join(hx(v['g2']) for v in vs)
    t = eng.g2_prepare(g2)
    out = eng.pairing_prepared(g1, t, with_final_exp=True)
    ml = eng.pairing_prepared(g1, t, with_final_exp=False)
    for i, v in enumerate(vs):
        assert out[576 * i:576 * (i + 1)] == hx(v['pairing']), i
        assert ml[576 * i:576 * (i + 1)] == hx(v['miller']), i
    # one prepared Q against every P (the reference's grouping of keys by message, index.ts:804-812)
    t0 = t[:eng.LINE_WIRE_BYTES]
    out = eng.pairing_prepared(g1, t0, with_final_exp=False)
    for i in range(n):
        assert out[576 * i:576 * (i + 1)] == oracle.miller_loop(g1[96 * i:96 * i + 96], g2[:192]), i
    # product over prepared tables == product of the separate Miller loops, with and without the shared final exponentiation
    for fe in (False, True):
        assert eng.pairing_prepared(g1, t, with_final_exp=fe, product=True) == oracle.miller_product(g1, g2, final_exp=fe)
        assert eng.pairing_prepared(g1, t0, with_final_exp=fe, product=True) == oracle.miller_product(g1, g2[:192] * n, final_exp=fe)
    assert eng.pairing_prepared(b'', b'', with_final_exp=True, product=True) == oracle.miller_product(b'', b'', final_exp=True)
