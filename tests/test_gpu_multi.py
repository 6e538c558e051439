"""Multi-GPU inside the library (nbls_init_multi): sharded pairings, Miller product and verifyBatch must be bit-equal
to the single-device result.  On a one-GPU box the handle is opened on device 0 several times (two and three contexts on one GPU), which
runs every line of the sharded path including the hipMemcpyPeer gather; with two or more devices the BASELINE configurations run across
real devices: configs[3] (a 2^17-pairing slice per GPU would take the oracle minutes, so equality is against the single-GPU engine) and
configs[4] (2^18-term product with one shared final exponentiation, constructed to equal ONE)."""
import hashlib
import importlib
import pytest
import torch
from goldenio import hx

pytestmark = pytest.mark.gpu
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


@pytest.fixture(scope='module')
def pkg():
    return importlib.import_module('noble-bls12-381_amd')


@pytest.fixture(scope='module')
def eng(pkg):
    return pkg.Engine(0)


def _points(eng, oracle, n, seed):
    ks = [int.from_bytes(hashlib.sha256(b'multi-%d-%d' % (seed, i)).digest(), 'big') % R or 1 for i in range(n)]
    P, st = eng.point_mul_batch([k.to_bytes(32, 'big') for k in ks]); assert not any(st)
    Q, st = eng.point_mul_batch([((k * 5 + 1) % R or 1).to_bytes(32, 'big') for k in ks], pts=oracle.g2_generator() * n, g2=True); assert not any(st)
    return P, Q


@pytest.mark.parametrize('devices', [[0, 0], [0, 0, 0]], ids=['2ctx', '3ctx'])
def test_sharded_paths_on_one_gpu(pkg, eng, oracle, golden, devices):
    m = pkg.MultiEngine(devices)
    assert m.n_devices == len(devices)
    for n in (1, 2, 7, 200):                      # fewer items than devices, ragged shards
        P, Q = _points(eng, oracle, n, n)
        for fe in (True, False):
            assert m.pairing_batch(P, Q, fe, False)[0] == eng.pairing_batch(P, Q, fe, False)[0], (n, fe)
            assert m.miller_product(P, Q, fe, False)[0] == oracle.miller_product(P, Q, final_exp=fe), (n, fe)
    assert m.miller_product(b'', b'', True)[0] == eng.miller_product(b'', b'', True)[0]
    # golden product with a known value (e(G1, G2)^41)
    p = golden['product']
    assert m.miller_product(hx(''.join(p['g1'])), hx(''.join(p['g2'])), True)[0] == hx(p['result'])
    # validation is per shard, the status array is the caller's
    P, Q = _points(eng, oracle, 5, 99)
    bad = P[:96 * 3] + bytes(95) + b'\x05' + P[96 * 4:]
    with pytest.raises(pkg.NblsError):
        m.miller_product(bad, Q, True, True)
    out, st = m.pairing_batch(bad, Q, True, True)
    ref, st1 = eng.pairing_batch(bad, Q, True, True)
    assert out == ref and st == st1 and st[3] != 0
    # verifyBatch: true, flipped message, identity key, undecodable key
    n = 301
    sks = [(int.from_bytes(hashlib.sha256(b'multi-sk' + i.to_bytes(2, 'big')).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'multi-msg' + i.to_bytes(2, 'big')).digest()[:1 + i % 32] for i in range(n)]     # ragged message lengths
    msgs = [m_ + bytes([i & 255, i >> 8]) for i, m_ in enumerate(msgs)]
    pks, sig = oracle.aggregate_sign(msgs, sks, threads=16)
    assert m.verify_batch(sig, msgs, pks) is True
    assert m.verify_batch(sig, msgs[:1], pks[:1]) is False or n == 1
    forged = list(msgs); forged[n - 1] = b'x'
    assert m.verify_batch(sig, forged, pks) is False
    ident = list(pks); ident[n // 2] = bytes([0xc0]) + bytes(47)
    assert m.verify_batch(sig, msgs, ident) is False
    undec = [hx(v['hex']) for v in golden['codec']['g1'] if v['result'] == 'Invalid compressed G1 point'][0]
    broken = list(pks); broken[n - 1] = undec
    with pytest.raises(pkg.NblsError, match='decode'):
        m.verify_batch(sig, msgs, broken)
    sig1 = oracle.sign(msgs[0], sks[0])[1]
    assert m.verify_batch(sig1, msgs[:1], pks[:1]) is True      # one signature, several devices: the surplus devices stay idle
    m.close()


def test_config3_full_size_eight_contexts(pkg, eng, oracle):
    """BASELINE configs[3] at its full size -- 2^20 independent pairings sharded eight ways -- in ONE nbls_multi_pairing_batch call.  On a
    one-GPU box the eight shards are eight contexts on device 0 (2^17 pairs each, the per-GPU share); with eight GPUs they are the real
    devices.  4096 distinct pairs x 256: the 4096 distinct results against the oracle, every repetition equal to the first (bit-exact,
    wherever in whichever shard the pair sits)."""
    G = torch.cuda.device_count()
    m = pkg.MultiEngine(None if G >= 8 else [i % G for i in range(8)])
    assert m.n_devices >= 8
    base_n, reps = 4096, 256
    P, Q = _points(eng, oracle, base_n, 20)
    out = m.pairing_batch(P * reps, Q * reps, True, False)[0]
    m.close()
    assert len(out) == 576 * base_n * reps
    first = out[:576 * base_n]
    ref, _ = oracle.pairing_batch(P, Q, True, False, threads=64)
    assert first == ref
    mv = memoryview(out)
    for r in range(1, reps):
        assert mv[576 * base_n * r:576 * base_n * (r + 1)] == first, r


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two or more GPUs')
def test_across_real_devices(pkg, eng, oracle):
    G = torch.cuda.device_count()
    m = pkg.MultiEngine(None)
    assert m.n_devices == G
    per = 1 << 17                                  # BASELINE configs[3]: 2^20 pairings over 8 GPUs = 2^17 per GPU
    base_n = 4096
    P, Q = _points(eng, oracle, base_n, 7)
    reps = per * G // base_n
    G1, G2 = P * reps, Q * reps
    out = m.pairing_batch(G1, G2, True, False)[0]
    ref = eng.pairing_batch(P, Q, True, False)[0]
    assert out == ref * reps
    # configs[4]: 2^18 terms, pairs (P, Q), (-P, Q) interleaved: the product is ONE
    half = 1 << 17
    P2, Q2 = P * (half // base_n), Q * (half // base_n)
    negP = b''.join(oracle.un('g1_neg_aff', P[96 * i:96 * i + 96], 96) for i in range(base_n)) * (half // base_n)
    one = (1).to_bytes(48, 'big') + bytes(528)
    assert m.miller_product(P2 + negP, Q2 + Q2, True)[0] == one
    assert m.miller_product(P2 + negP, Q2 + Q2, True)[0] == eng.miller_product(P2 + negP, Q2 + Q2, True)[0]
    # did the 576-byte partials travel peer to peer (xGMI) or staged by the runtime?  Reported either way; on a node whose devices are peers it must be peer to peer
    peers = m.peer_access()
    can = [torch.cuda.can_device_access_peer(0, g) for g in range(1, G)]
    print('peer access to the reducing device:', peers, 'torch says', can)
    assert peers[0] == 1 and all(p == 1 for p, c in zip(peers[1:], can) if c)
    m.close()
    # a reducing device other than device 0: the LAST device leads, the others follow in reverse order
    order = list(range(G - 1, -1, -1))
    m2 = pkg.MultiEngine(order)
    k = 2048 * G
    sub1, sub2 = P2[:96 * k] + negP[:96 * k], Q2[:192 * k] + Q2[:192 * k]
    assert m2.miller_product(sub1, sub2, True)[0] == one
    vs = 512 * G
    assert m2.pairing_batch(G1[:96 * vs], G2[:192 * vs], True, False)[0] == (ref * reps)[:576 * vs]
    assert all(p in (0, 1) for p in m2.peer_access())
    m2.close()
