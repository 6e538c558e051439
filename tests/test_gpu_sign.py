"""GPU parity tests for the secret-scalar side: getPublicKey / sign (reference index.ts:738-752) through the
C ABI (nbls_g1_mul_batch, nbls_g2_mul_batch, nbls_sign_batch) against the reference's own sign vectors
(test/bls12-381/fixtures/bls12-381-g2-test-vectors.txt, repacked in tests/golden/ref_testdata.json.gz), reference-generated
key/signature triples and the oracle."""
import importlib
import random
import pytest
from goldenio import hx

pytestmark = pytest.mark.gpu
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


@pytest.fixture(scope='module')
def eng():
    return importlib.import_module('noble-bls12-381_amd').Engine(0)


def test_sign_vectors(eng, testdata):
    vs = testdata['sign_vectors']
    assert len(vs) == 559
    sigs = eng.sign_batch([hx(v[1]) for v in vs], [hx(v[0]) for v in vs])
    for v, s in zip(vs, sigs):
        assert s == hx(v[2]), v[0]


def test_public_keys_and_signatures_of_reference_run(eng, golden):
    vs = golden['sigs']
    assert eng.get_public_keys([hx(v['sk']) for v in vs]) == [hx(v['pk']) for v in vs]
    assert eng.sign_batch([hx(v['msg']) for v in vs], [hx(v['sk']) for v in vs]) == [hx(v['sig']) for v in vs]


def test_point_mul_structured_scalars(eng, oracle):
    rnd = random.Random(381)
    ks = [1, 2, 3, R - 1, R + 1, 2 * R + 5, (1 << 256) - 1, 1 << 255, 1 << 200, (1 << 128) - 1] + [rnd.randrange(1, R) for _ in range(54)]
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    p1 = [oracle.g1_mul(g1, rnd.randrange(1, R))[1] for _ in range(4)]
    p2 = [oracle.g2_mul(g2, rnd.randrange(1, R))[1] for _ in range(4)]
    sc = [k.to_bytes(32, 'big') for k in ks]
    out, st = eng.point_mul_batch(sc)                      # generator
    assert st == bytes(len(ks))
    for i, k in enumerate(ks):
        assert out[96 * i:96 * i + 96] == oracle.g1_mul(g1, k % R)[1]
    pts = b''.join(p1[i % 4] for i in range(len(ks)))
    out, st = eng.point_mul_batch(sc, pts)
    for i, k in enumerate(ks):
        assert out[96 * i:96 * i + 96] == oracle.g1_mul(p1[i % 4], k % R)[1]
    pts = b''.join(p2[i % 4] for i in range(len(ks)))
    out, st = eng.point_mul_batch(sc, pts, g2=True)
    assert st == bytes(len(ks))
    for i, k in enumerate(ks):
        assert out[192 * i:192 * i + 192] == oracle.g2_mul(p2[i % 4], k % R)[1]


def test_invalid_keys(eng):
    """normalizePrivKey (index.ts:269-279): keys that are 0 mod r are rejected"""
    pkg = importlib.import_module('noble-bls12-381_amd')
    for bad in (0, R, 2 * R):
        out, st = eng.point_mul_batch([bad.to_bytes(32, 'big'), (5).to_bytes(32, 'big')])
        assert st == bytes([5, 0])
        with pytest.raises(pkg.NblsError):
            eng.get_public_keys([bad.to_bytes(32, 'big')])
        with pytest.raises(pkg.NblsError):
            eng.sign_batch([b'msg'], [bad.to_bytes(32, 'big')])


def test_sign_then_verify_batch(eng):
    """signatures produced on the GPU verify on the GPU: aggregate of distinct-message signatures through verifyBatch"""
    rnd = random.Random(7)
    n = 33
    sks = [rnd.randrange(1, R).to_bytes(32, 'big') for _ in range(n)]
    msgs = [bytes([i]) * (i + 1) for i in range(n)]
    pks = eng.get_public_keys(sks)
    sigs = eng.sign_batch(msgs, sks)
    affs, st = eng.decompress_batch(b''.join(sigs), True)
    assert st == [0] * n
    agg, z = eng.point_sum(affs, g2=True)
    assert z == 0
    assert eng.verify_batch(eng.compress_g2(agg), msgs, pks) is True
    assert eng.verify_batch(eng.compress_g2(agg), msgs[::-1], pks) is False


def test_sign_with_everything_resident_in_hbm(eng, testdata):
    """nbls_sign_batch_dev (round 5): messages, offsets and keys in device memory -> affine points and status bytes in device memory, against the reference's 559 sign vectors
    (ragged message lengths, the empty message among them); a key that is 0 mod r gives the zero point and status 1; decreasing offsets are refused; staging buffers are reused
    between host-buffer calls of different sizes (the pooled HostIO) without mixing results up"""
    import numpy as np
    import torch
    pkg = importlib.import_module('noble-bls12-381_amd')
    vs = testdata['sign_vectors']
    msgs = [hx(v[1]) for v in vs] + [b'', b'zero-key']
    keys = [hx(v[0]) for v in vs] + [(7).to_bytes(32, 'big'), R.to_bytes(32, 'big')]
    n = len(msgs)
    offs = np.zeros(n + 1, dtype=np.uint32); offs[1:] = np.cumsum([len(m) for m in msgs])
    d_msgs = torch.frombuffer(bytearray(b''.join(msgs) + b'\0'), dtype=torch.uint8).cuda()
    d_offs = torch.from_numpy(offs.view(np.int32)).cuda()
    d_keys = torch.frombuffer(bytearray(b''.join(keys)), dtype=torch.uint8).cuda()
    d_out = torch.empty(192 * n, dtype=torch.uint8, device='cuda'); d_st = torch.full((n,), 77, dtype=torch.uint8, device='cuda')
    eng.sign_batch_dev(n, d_msgs.data_ptr(), d_offs.data_ptr(), d_keys.data_ptr(), d_out.data_ptr(), d_st.data_ptr())
    out = bytes(d_out.cpu().numpy().tobytes()); st = bytes(d_st.cpu().numpy().tobytes())
    assert st == bytes(n - 1) + bytes([1])
    sigs = eng.compress_batch(out[:192 * (n - 1)], g2=True)
    for i, v in enumerate(vs):
        assert sigs[96 * i:96 * i + 96] == hx(v[2]), i
    assert sigs[96 * (n - 2):96 * (n - 1)] == eng.sign_batch([b''], [(7).to_bytes(32, 'big')])[0]
    # a second stream, and a sub-range of the same buffers
    s2 = torch.cuda.Stream()
    d_out2 = torch.empty(192 * 100, dtype=torch.uint8, device='cuda'); d_st2 = torch.empty(100, dtype=torch.uint8, device='cuda')
    eng.sign_batch_dev(100, d_msgs.data_ptr(), d_offs.data_ptr(), d_keys.data_ptr(), d_out2.data_ptr(), d_st2.data_ptr(), stream=s2.cuda_stream)
    assert bytes(d_out2.cpu().numpy().tobytes()) == out[:192 * 100]
    bad = offs.copy(); bad[5] = bad[6] + 3
    d_bad = torch.from_numpy(bad.view(np.int32)).cuda()
    with pytest.raises(pkg.NblsError):
        eng.sign_batch_dev(n, d_msgs.data_ptr(), d_bad.data_ptr(), d_keys.data_ptr(), d_out.data_ptr(), d_st.data_ptr())
    # pooled staging buffers: host-buffer calls of changing sizes interleaved, every result as before
    ref = [hx(v[2]) for v in vs]
    for k in (3, 200, 17, 559, 64, 1):
        assert eng.sign_batch([hx(v[1]) for v in vs[:k]], [hx(v[0]) for v in vs[:k]]) == ref[:k]
        assert eng.get_public_keys([hx(v[0]) for v in vs[:k]])[0] == eng.get_public_keys([hx(vs[0][0])])[0]


def test_both_ladders_of_sign(eng, testdata, oracle):
    """sign's two ladders for points of G2 (csrc/pipelines_codec.cpp dev_point_mul): the sign-aligned one-addition-per-bit form (up to 6144 keys) and the windowed psi-split form (above) on the
    reference's 559 sign vectors and on structured keys -- even / odd low digit, digits rolling over at powers of |z|, r - 1, r + 1, 2^256 - 1 -- against the oracle"""
    vs = testdata['sign_vectors']
    Z = 0xd201000000010000
    ks = [1, 2, 3, 4, Z - 1, Z, Z + 1, Z * Z - 1, Z * Z, Z ** 3, Z ** 3 - 1, R - 1, R + 1, (1 << 256) - 1, (1 << 256) - 2, (1 << 255) + 12345]
    msgs = [b'structured-%d' % i for i in range(len(ks))]
    want = [oracle.sign(m, (k % R).to_bytes(32, 'big'))[1] for m, k in zip(msgs, ks)]
    hv = [oracle.hash_to_g2(m)[1] for m in msgs]
    try:
        for sac_max, ls2_max in ((0, 0), (6144, 0), (6144, 4096), (0, 4096)):     # the two-lane forms of the point chains (clearCofactor's ladders, the sign-aligned ladder) on and off
            eng.set_sac_max(sac_max); eng.set_pt_ls2_max(ls2_max)
            sigs = eng.sign_batch([hx(v[1]) for v in vs], [hx(v[0]) for v in vs])
            assert sigs == [hx(v[2]) for v in vs], (sac_max, ls2_max)
            assert eng.sign_batch(msgs, [k.to_bytes(32, 'big') for k in ks]) == want, (sac_max, ls2_max)
            assert eng.hash_to_g2_batch(msgs) == b''.join(hv), (sac_max, ls2_max)
    finally:
        eng.set_sac_max(6144); eng.set_pt_ls2_max(4096)


def test_sign_dispatch_thresholds(eng, oracle):
    """batch sizes on both sides of the two dispatch thresholds of sign (two-lane point chains up to 4096 items, sign-aligned ladder with a projective hand-over up to 6144 keys,
    windowed ladder on affine points above): every size gives the signatures of the small-batch path, which is pinned on the reference's vectors above; spot checks against the oracle"""
    import hashlib
    n = 6146
    sks = [(int.from_bytes(hashlib.sha256(b'thr-sk%d' % i).digest(), 'big') % (R - 1) + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'thr-m%d' % i).digest()[:1 + i % 32] for i in range(n)]
    ref = []
    for i in range(0, n, 512):
        ref += eng.sign_batch(msgs[i:i + 512], sks[i:i + 512])
    for i in (0, 4095, 4096, 6143, 6144, 6145):
        assert ref[i] == oracle.sign(msgs[i], sks[i])[1], i
    for m in (4096, 4097, 6144, 6145, 6146):
        assert eng.sign_batch(msgs[:m], sks[:m]) == ref[:m], m


def test_hash_to_g2_dispatch_threshold(eng, oracle):
    """hash-to-G2 on both sides of the size up to which clearCofactor's ladders run in their two-lane forms (4096 messages): equal to the chunked small-batch results, which the
    RFC 9380 / reference vectors pin (tests/test_gpu_reference_vectors.py); spot checks against the oracle"""
    import hashlib
    n = 4098
    msgs = [hashlib.sha256(b'thr-h%d' % i).digest()[:1 + i % 32] for i in range(n)]
    ref = b''.join(eng.hash_to_g2_batch(msgs[i:i + 600]) for i in range(0, n, 600))
    for i in (0, 4095, 4096, 4097):
        assert ref[192 * i:192 * i + 192] == oracle.hash_to_g2(msgs[i])[1], i
    for m in (4096, 4097, 4098):
        assert eng.hash_to_g2_batch(msgs[:m]) == ref[:192 * m], m


def test_hash_to_g2_norm_method_threshold(eng, oracle, testdata):
    """round 6: from 32,768 messages hash-to-G2 takes the square root of its SWU maps by the norm method (P_H2C_NA / NM / NB around two Fp exponentiations, csrc/codec.h swu_norm_*)
    instead of the reference's one Fp2 exponentiation (math.ts:1195-1214) -- the same points, since map_to_curve fixes the root's sign itself (math.ts:1264).  Both sides of the
    switch, each method forced at every size, the RFC 9380 vectors of test/hashToCurve.test.ts through the norm method, the oracle on a strided sample."""
    import hashlib
    from goldenio import hx
    msgs = [hashlib.sha256(b'norm-h%d' % i).digest()[:1 + i % 40] for i in range(2100)]
    try:
        eng.set_h2c_norm_min(1 << 30); fp2 = eng.hash_to_g2_batch(msgs)
        eng.set_h2c_norm_min(0); nrm = eng.hash_to_g2_batch(msgs)
        assert nrm == fp2
        for m in (1, 2, 3, 65):
            assert eng.hash_to_g2_batch(msgs[:m]) == fp2[:192 * m], m
        suite = testdata['h2c_g2_ro']
        out = eng.hash_to_g2_batch([hx(v['msg']) for v in suite['vectors']], suite['dst'].encode())
        for i, v in enumerate(suite['vectors']):
            e = hx(v['x1x0y1y0'])
            assert out[192 * i:192 * (i + 1)] == e[48:96] + e[0:48] + e[144:192] + e[96:144]
        eng.set_h2c_norm_min(2048)
        for m in (2047, 2048, 2049):
            assert eng.hash_to_g2_batch(msgs[:m]) == fp2[:192 * m], m
        assert eng.program_kernel('h2c_na') == 'nbls_aot_h2c_n' and eng.program_kernel('h2c_nb') == 'nbls_aot_h2c_n'
    finally:
        eng.set_h2c_norm_min(32768)
    for i in list(range(0, 2100, 97)) + [2047, 2048, 2099]:
        assert fp2[192 * i:192 * i + 192] == oracle.hash_to_g2(msgs[i])[1], i


def test_wide_power_kernel_threshold(eng, oracle):
    """round 6: launches of at most 3072 elements run the fixed-exponent powers in their one-limb-per-lane form (nbls_pow_wide_kernel, csrc/pow_wide.h) -- the Fp2 exponent of
    hash-to-G2 (two elements per message: 1536 / 1537 messages straddle the switch), the Fp2 square root of a compressed signature and the Fp square root of a compressed key (3072 /
    3073 encodings): both sides against the oracle, item by item on a strided sample and in full against each other"""
    import hashlib
    msgs = [hashlib.sha256(b'wide-h%d' % i).digest()[:1 + i % 32] for i in range(1545)]
    small = eng.hash_to_g2_batch(msgs[:1536]); big = eng.hash_to_g2_batch(msgs)
    assert big[:192 * 1536] == small
    for i in list(range(0, 1545, 97)) + [1535, 1536, 1537, 1544]:
        assert big[192 * i:192 * i + 192] == oracle.hash_to_g2(msgs[i])[1], i
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    ks = [(int.from_bytes(hashlib.sha256(b'wide-k%d' % i).digest(), 'big') % (1 << 250) + 1).to_bytes(32, 'big') for i in range(3080)]
    P1, st = eng.point_mul_batch(ks); assert not any(st)
    P2, st = eng.point_mul_batch(ks, pts=g2 * len(ks), g2=True); assert not any(st)
    for is_g2, aff, a in ((False, P1, 96), (True, P2, 192)):
        comp = eng.compress_batch(aff, g2=is_g2)
        bad = bytearray(comp[:a // 2]); bad[-1] ^= 3      # one encoding without a square root or outside the subgroup among them
        comp = bytes(bad) + comp[a // 2:]
        for m in (1, 2, 3072, 3073, 3080):
            out, st = eng.decompress_batch(comp[:a // 2 * m], g2=is_g2)
            ro, rs = oracle.decompress_batch(comp[:a // 2 * m], is_g2, 32)
            assert bytes(st) == rs and out == ro, (is_g2, m)
