"""Differential soak on the box (round-5 review item 7; the reference's own decoder / verify tests: test/index.test.ts:37-261, 337-398), sized for ~90 s:
  * 10^6 random and malformed point encodings -- every combination of the three flag bits x valid / random / boundary coordinates (0, p - 1, p, p + 1, 2^381 - 1, values above p)
    -- through nbls_g1_decompress_batch (PointG1.fromHex, 48 B) and nbls_g2_decompress_batch (PointG2.fromSignature, 96 B): EVERY status byte and every decoded point against the
    oracle (threaded C restatement), plus equality of the status histograms;
  * 300 verifyBatch calls of random sizes with random corruptions against the oracle's verdict (true / false / throws);
  * 2,000 pool steps of random sizes 1 .. 8192 on twelve contexts: every batch compared in full with the oracle's results for the same pairs, free device memory before and
    after (no growth)."""
import ctypes as C
import hashlib
import importlib
import os
import random
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
P_INT = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
THREADS = min(128, os.cpu_count() or 8)


@pytest.fixture(scope='module')
def eng():
    pkg = importlib.import_module('noble-bls12-381_amd')
    return pkg.Engine(0)


def _mutate(base, n, a, seed):
    """n encodings of `a` bytes from the valid compressed points in `base` (k x a uint8): category i % 16 decides what happens to item i"""
    rng = np.random.default_rng(seed)
    k = base.shape[0]
    x = base[np.arange(n) % k].copy()
    cat = np.arange(n) % 16
    idx = np.arange(n)
    # 6: one random bit of the encoding flipped (mostly: no square root / not in the subgroup)
    m = idx[cat == 6]; byte = rng.integers(0, a, m.size); bit = rng.integers(0, 8, m.size); x[m, byte] ^= (1 << bit).astype(np.uint8)
    # 7: random bytes, all eight flag combinations
    m = idx[cat == 7]; x[m] = rng.integers(0, 256, (m.size, a), dtype=np.uint8); x[m, 0] = (x[m, 0] & 0x1f) | ((np.arange(m.size) % 8) << 5).astype(np.uint8)
    # 8: a valid point under every flag combination
    m = idx[cat == 8]; x[m, 0] = (x[m, 0] & 0x1f) | ((np.arange(m.size) % 8) << 5).astype(np.uint8)
    # 9: x = 0 under every flag combination
    m = idx[cat == 9]; x[m] = 0; x[m, 0] = ((np.arange(m.size) % 8) << 5).astype(np.uint8)
    # 10: boundary coordinates in the (first) 48-byte word: p - 1, p, p + 1, 2^381 - 1, 1, 2, p + 2^380 mod 2^381 ...
    m = idx[cat == 10]
    bnd = [P_INT - 1, P_INT, P_INT + 1, (1 << 381) - 1, 1, 2, 3, 4, P_INT - 2, (1 << 380), P_INT + 5, (1 << 381) - 2]
    for j, i in enumerate(m):
        v = bnd[j % len(bnd)]; x[i, :48] = np.frombuffer(v.to_bytes(48, 'big'), dtype=np.uint8); x[i, 0] |= 0x80 | (0x20 if (j // len(bnd)) & 1 else 0)
        if a == 96 and (j // (2 * len(bnd))) & 1: x[i, 48:] = np.frombuffer(bnd[(j + 3) % len(bnd)].to_bytes(48, 'big'), dtype=np.uint8)
    # 11: infinity flag over non-zero bytes; 12: all zero; 13: the canonical infinity encoding
    m = idx[cat == 11]; x[m, 0] |= 0x40
    m = idx[cat == 12]; x[m] = 0
    m = idx[cat == 13]; x[m] = 0; x[m, 0] = 0xc0
    # 14: random coordinates with the compression bit (x >= p included: the reference reduces)
    m = idx[cat == 14]; x[m] = rng.integers(0, 256, (m.size, a), dtype=np.uint8); x[m, 0] |= 0x80; x[m, 0] &= 0xbf
    # 15: the sort bit flipped on a valid point (the other root: valid)
    m = idx[cat == 15]; x[m, 0] ^= 0x20
    return x


def test_decode_soak_one_million(eng, oracle):
    rnd = random.Random(6)
    ks = [rnd.randrange(1, 1 << 250).to_bytes(32, 'big') for _ in range(256)]
    g1, st = eng.point_mul_batch(ks); assert not any(st)
    g2, st = eng.point_mul_batch(ks[:96], pts=oracle.g2_generator() * 96, g2=True); assert not any(st)
    assert g1[:96] == oracle.g1_mul(oracle.g1_generator(), int.from_bytes(ks[0], 'big'))[1]
    c1 = np.frombuffer(eng.compress_batch(g1, g2=False), dtype=np.uint8).reshape(256, 48)
    c2 = np.frombuffer(eng.compress_batch(g2, g2=True), dtype=np.uint8).reshape(96, 96)
    total_bad = 0
    for is_g2, base, n, a in ((False, c1, 900_000, 48), (True, c2, 100_000, 96)):
        enc = _mutate(base, n, a, 1234 + a).tobytes()
        out, st = eng.decompress_batch(enc, g2=is_g2)
        ref_out, ref_st = oracle.decompress_batch(enc, is_g2, THREADS)
        st = np.asarray(st, dtype=np.int8); rs = np.frombuffer(ref_st, dtype=np.int8)
        diff = np.nonzero(st != rs)[0]
        assert diff.size == 0, ('status', is_g2, int(diff[0]), int(st[diff[0]]), int(rs[diff[0]]), enc[a * int(diff[0]):a * int(diff[0]) + a].hex())
        assert np.array_equal(np.bincount(st.astype(np.int64) + 1, minlength=12), np.bincount(rs.astype(np.int64) + 1, minlength=12))
        go = np.frombuffer(out, dtype=np.uint8).reshape(n, 2 * a); ro = np.frombuffer(ref_out, dtype=np.uint8).reshape(n, 2 * a)
        bad = np.nonzero((go != ro).any(axis=1))[0]
        assert bad.size == 0, ('point', is_g2, int(bad[0]))
        hist = {int(v): int(c) for v, c in zip(*np.unique(rs, return_counts=True))}
        print('soak decode %s: %d encodings, statuses %s' % ('G2' if is_g2 else 'G1', n, hist))
        assert hist.get(0, 0) > n // 3 and len(hist) >= 4      # valid, zero, not in subgroup, no square root all occur
        total_bad += n - hist.get(0, 0)
    assert total_bad > 100_000


def test_verify_batch_soak(eng, oracle):
    rnd = random.Random(66)
    ns = 48
    sks = [(int.from_bytes(hashlib.sha256(b'soak-sk%d' % i).digest(), 'big') % (1 << 254) + 1).to_bytes(32, 'big') for i in range(ns)]
    pks = [oracle.get_public_key(s) for s in sks]
    outcomes = {}
    for it in range(300):
        n = rnd.randrange(1, 34)
        who = [rnd.randrange(ns) for _ in range(n)]
        msgs = [hashlib.sha256(b'soak-m-%d-%d' % (it, j)).digest()[:1 + (it + j) % 32] for j in range(n)]
        p, agg = oracle.aggregate_sign(msgs, [sks[w] for w in who], threads=8)
        assert p == [pks[w] for w in who]
        kind = it % 8
        if kind == 1: j = rnd.randrange(n); msgs[j] = msgs[j] + b'!'
        elif kind == 2 and n > 1: p[0], p[-1] = p[-1], p[0]
        elif kind == 3: j = rnd.randrange(n); b = bytearray(p[j]); b[rnd.randrange(1, 48)] ^= 1 << rnd.randrange(8); p[j] = bytes(b)      # mostly: no root / not in the subgroup -> throws
        elif kind == 4: p[rnd.randrange(n)] = bytes([0xc0]) + bytes(47)           # zero key: pairing throws inside the try block -> false
        elif kind == 5: b = bytearray(agg); b[rnd.randrange(1, 96)] ^= 1 << rnd.randrange(8); agg = bytes(b)
        elif kind == 6: agg = bytes([0xc0]) + bytes(95)
        expect = oracle.verify_batch(agg, msgs, p)      # 1 / 0 / negative: the reference throws
        try:
            got = 1 if eng.verify_batch(agg, msgs, p) else 0
        except Exception:      # noqa: BLE001 -- NblsError: the reference throws while decoding
            got = -1
        assert got == (expect if expect >= 0 else -1), (it, kind, n, expect, got)
        outcomes[got] = outcomes.get(got, 0) + 1
    print('soak verifyBatch outcomes', outcomes)
    assert set(outcomes) == {1, 0, -1}


def test_pool_soak_random_sizes_no_growth(oracle):
    pkg = importlib.import_module('noble-bls12-381_amd')
    import bench
    D, NMAX, STEPS = 12, 8192, 2000
    pipe = pkg.PairingPipeline(0, D)
    G1, G2 = bench.synth_stream(pipe.engines[0], oracle, NMAX, seed=0x736f616b)
    ref, _ = oracle.pairing_batch(G1, G2, True, False, threads=THREADS)
    ref = np.frombuffer(ref, dtype=np.uint8).reshape(NMAX, 576)
    d1 = torch.frombuffer(bytearray(G1), dtype=torch.uint8).cuda(); d2 = torch.frombuffer(bytearray(G2), dtype=torch.uint8).cuda()
    outs = [torch.zeros(576 * NMAX, dtype=torch.uint8, device='cuda') for _ in range(D)]
    host = [torch.empty(576 * NMAX, dtype=torch.uint8).pin_memory() for _ in range(D)]
    for _ in range(D):      # every context grows its scratch to the largest size once
        pipe.submit(NMAX, d1.data_ptr(), d2.data_ptr(), outs[pipe.slot].data_ptr(), True)
    torch.cuda.synchronize()
    free0, _total = torch.cuda.mem_get_info()
    rnd = random.Random(2026)
    pending = [None] * D      # (offset, n) of the batch last submitted to a slot
    checked = 0

    def check(slot):
        nonlocal checked
        off, n = pending[slot]
        got = host[slot][:576 * n].numpy().reshape(n, 576)
        bad = np.nonzero((got != ref[off:off + n]).any(axis=1))[0]
        assert bad.size == 0, ('pool soak', slot, off, n, int(bad[0]))
        checked += n

    for step in range(STEPS):
        slot = pipe.slot
        if pending[slot] is not None:      # the slot's previous batch: wait for everything (round-robin: it is the oldest in flight), copy back, compare in full
            torch.cuda.synchronize()
            for s_ in range(D):
                if pending[s_] is not None:
                    host[s_][:576 * pending[s_][1]].copy_(outs[s_][:576 * pending[s_][1]])
                    check(s_); pending[s_] = None
        n = rnd.choice((1, 2, 3, 63, 64, 65, 1024, 1025, 2048, 2049, 4096, 4097, NMAX)) if step % 5 == 0 else rnd.randrange(1, NMAX + 1)
        off = rnd.randrange(0, NMAX - n + 1)
        pipe.submit(n, d1.data_ptr() + 96 * off, d2.data_ptr() + 192 * off, outs[slot].data_ptr(), True)
        pending[slot] = (off, n)
    torch.cuda.synchronize()
    for s_ in range(D):
        if pending[s_] is not None:
            host[s_][:576 * pending[s_][1]].copy_(outs[s_][:576 * pending[s_][1]]); check(s_)
    free1, _total = torch.cuda.mem_get_info()
    print('pool soak: %d steps, %d pairings compared in full, free memory %d -> %d MiB' % (STEPS, checked, free0 >> 20, free1 >> 20))
    assert free0 - free1 < (64 << 20), (free0, free1)      # no growth (the allocator's own rounding aside)
