"""GPU tests of the multi-GPU building blocks on one GPU: nbls_verify_batch_partial_dev + nbls_fp12_product_final_dev through
parallel.verify_batch_sharded (the collectives themselves are covered by the world-size-2 gloo test on CPU)."""
import importlib

import pytest
import torch

from goldenio import hx
import oracle_py

pytestmark = pytest.mark.gpu


def _dev(b):
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()


def test_verify_batch_sharded_one_rank(oracle, golden):
    pkg = importlib.import_module('noble-bls12-381_amd')
    par = importlib.import_module('noble-bls12-381_amd.parallel')
    eng = pkg.Engine(0)
    be = par.EngineBackend(eng)
    vb = golden['verify_batch']
    sig, msgs, pks = hx(vb['agg_sig']), [hx(m) for m in vb['msgs']], [hx(p) for p in vb['pks']]
    uni = [oracle.expand_message_xmd(m, oracle_py.DST_DEFAULT, 256) for m in msgs]
    d_sig, d_uni, d_pk = _dev(sig), _dev(b''.join(uni)), _dev(b''.join(pks))
    assert par.verify_batch_sharded(be, d_sig, d_uni, d_pk) is True
    bad = list(uni); bad[1] = oracle.expand_message_xmd(b'some other message', oracle_py.DST_DEFAULT, 256)
    assert par.verify_batch_sharded(be, d_sig, _dev(b''.join(bad)), d_pk) is False
    # two "ranks" by hand: shard 0 carries the signature pair, shard 1 does not; partials multiplied by the finishing call
    n = len(msgs); h = n // 2
    p0, z0 = be.verify_partial(d_sig, _dev(b''.join(uni[:h])), _dev(b''.join(pks[:h])))
    p1, z1 = be.verify_partial(None, _dev(b''.join(uni[h:])), _dev(b''.join(pks[h:])))
    assert not z0 and not z1
    res = be.finish(torch.cat([p0, p1]), True)
    assert bytes(res.cpu().numpy().tobytes()) == par.ONE_FP12
    # the partial of shard 1 is the plain Miller product of its pairs (oracle: decompress, hash, multiply, no final exponentiation)
    g1 = b''.join(oracle.call('g1_decompress', 96, pk)[1] for pk in pks[h:]); g2 = b''.join(oracle.hash_to_g2(m)[1] for m in msgs[h:])
    assert bytes(p1.cpu().numpy().tobytes()) == oracle.miller_product(g1, g2, False)
    # an undecodable key raises as the reference does (PointG1.fromHex: x^3 + 4 is not a square)
    broken = bytearray(pks[0])
    while oracle.call('g1_decompress', 96, bytes(broken))[0] != 4:
        broken[47] = (broken[47] + 1) & 0xff
    with pytest.raises(pkg.NblsError):
        par.verify_batch_sharded(be, d_sig, d_uni, _dev(bytes(broken) + b''.join(pks[1:])))
