#!/usr/bin/env python3
"""Round 6 experiment: nbls_sign_batch on n keys as ONE call against TWO concurrent calls of n / 2 on two contexts (host threads): does the latency-bound chain of a medium
batch overlap with itself?  Host buffers, C ABI only."""
import ctypes as C, hashlib, importlib, os, statistics, sys, threading, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '22')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
pkg = importlib.import_module('noble-bls12-381_amd')
engs = [pkg.Engine(0) for _ in range(4)]
for n in (4096, 8192, 16384, 32768):
    sks = [(int.from_bytes(hashlib.sha256(b'ab-sk' + i.to_bytes(4, 'big')).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'ab-m' + i.to_bytes(4, 'big')).digest() for i in range(n)]
    def pack(lo, hi):
        blob = b''.join(msgs[lo:hi]); offs_np = np.zeros(hi - lo + 1, dtype=np.uint32); offs_np[1:] = np.cumsum([len(m) for m in msgs[lo:hi]])
        return blob, (C.c_uint32 * (hi - lo + 1)).from_buffer(offs_np), b''.join(sks[lo:hi]), C.create_string_buffer(192 * (hi - lo)), C.create_string_buffer(hi - lo), offs_np
    res = {}
    for parts in (1, 2, 4):
        cuts = [n * k // parts for k in range(parts + 1)]
        packs = [pack(cuts[k], cuts[k + 1]) for k in range(parts)]
        def call(k): p = packs[k]; engs[k].sign_packed(cuts[k + 1] - cuts[k], p[0], p[1], p[2], p[3], p[4])
        def run():
            th = [threading.Thread(target=call, args=(k,)) for k in range(1, parts)]
            for t in th: t.start()
            call(0)
            for t in th: t.join()
        for _ in range(3): run()
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
        res[parts] = round(statistics.median(ts), 3)
        if parts == 1: ref = packs[0][3].raw
        else: assert b''.join(p[3].raw for p in packs) == ref
    print('SIGN_SPLIT n=%d ms:' % n, res, 'sigs/s: %s' % {k: round(n / v / 1e3, 3) for k, v in res.items()}, flush=True)
