"""world_size-2 gloo test of the sharded multi-pairing product on CPU: the collective logic of
parallel.miller_product_sharded with the oracle standing in for the GPU engine."""
import importlib
import os
import sys
import hashlib
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    def __init__(self, oracle):
        self.o = oracle

    def local_product(self, g1, g2):
        return torch.frombuffer(bytearray(self.o.miller_product(bytes(g1.numpy().tobytes()), bytes(g2.numpy().tobytes()), False)), dtype=torch.uint8)

    def verify_partial(self, sig96, msgs, pks48):
        """msgs: list of message byte strings, pks48: list of compressed keys, sig96: bytes or None (the oracle hashes the messages itself)"""
        g1 = [self.o.call('g1_decompress', 96, pk)[1] for pk in pks48]
        g2 = [self.o.hash_to_g2(m)[1] for m in msgs]
        if sig96 is not None:
            g1.append(self.o.un('g1_neg_aff', self.o.g1_generator(), 96)); g2.append(self.o.call('g2_decompress', 192, sig96)[1])
        return torch.frombuffer(bytearray(self.o.miller_product(b''.join(g1), b''.join(g2), False)), dtype=torch.uint8), False

    def finish(self, partials, final_exp=True):
        import ctypes as C
        raw = bytes(partials.numpy().tobytes())
        acc = raw[:576]
        for k in range(1, len(raw) // 576):
            acc = self.o.bin('fp12_mul', acc, raw[576 * k:576 * (k + 1)], 576)
        if final_exp:
            acc = self.o.un('fp12_final_exp', acc, 576)
        return torch.frombuffer(bytearray(acc), dtype=torch.uint8)


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle_py
    par = importlib.import_module('noble-bls12-381_amd.parallel')
    o = oracle_py.load(rebuild=False)
    g1g, g2g = o.g1_generator(), o.g2_generator()
    G1 = b''.join(o.g1_mul(g1g, int.from_bytes(hashlib.sha256(b'a%d' % i).digest(), 'big') % 2**200 + 1)[1] for i in range(n))
    G2 = b''.join(o.g2_mul(g2g, int.from_bytes(hashlib.sha256(b'b%d' % i).digest(), 'big') % 2**200 + 1)[1] for i in range(n))
    lo, hi = par.shard_bounds(n, world, rank)
    t1 = torch.frombuffer(bytearray(G1[96 * lo:96 * hi]), dtype=torch.uint8)
    t2 = torch.frombuffer(bytearray(G2[192 * lo:192 * hi]), dtype=torch.uint8)
    res = par.miller_product_sharded(OracleBackend(o), t1, t2, final_exp=True)
    ref = o.miller_product(G1, G2, True)
    q.put((rank, bytes(res.numpy().tobytes()) == ref))
    dist.barrier(); dist.destroy_process_group()


def _verify_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle_py
    par = importlib.import_module('noble-bls12-381_amd.parallel')
    o = oracle_py.load(rebuild=False)
    sks = [(int.from_bytes(hashlib.sha256(b'sk%d' % i).digest(), 'big') % 2**250 + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'msg%d' % i).digest() for i in range(n)]
    pks = [o.get_public_key(sk) for sk in sks]
    sig = o.aggregate_sign(msgs, sks, threads=2)[1]
    lo, hi = par.shard_bounds(n, world, rank)
    good = par.verify_batch_sharded(OracleBackend(o), sig, msgs[lo:hi], pks[lo:hi])
    wrong = list(msgs); wrong[n - 1] = b'another message'          # lives in the last rank's shard only: every rank must still answer False
    bad = par.verify_batch_sharded(OracleBackend(o), sig, wrong[lo:hi], pks[lo:hi])
    q.put((rank, good, bad, bool(o.verify_batch(sig, msgs, pks))))
    dist.barrier(); dist.destroy_process_group()


def test_sharded_verify_batch_gloo_world2(oracle):
    """verifyBatch with the (key, message) pairs sharded over two ranks: one all-gather of Fp12 partials, the signature pair on rank 0"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=_verify_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True, False, True), (1, True, False, True)]


def test_sharded_product_gloo_world2(oracle):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_bounds():
    par = importlib.import_module('noble-bls12-381_amd.parallel')
    for n in (0, 1, 7, 4096, 1000000):
        for w in (1, 2, 4, 8):
            b = [par.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
