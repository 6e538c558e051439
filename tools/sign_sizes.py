#!/usr/bin/env python3
"""sign / getPublicKey from host buffers against the batch size (one call at a time): at 8192 keys every kernel of the chain runs at one wavefront per SIMD, at 65,536 the
launches are dense.  usage: tools/sign_sizes.py [n,n,..]"""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # before the engine: torch brings its own HIP runtime
pkg = importlib.import_module('noble-bls12-381_amd')
eng = pkg.Engine(0)
sizes = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [2048, 8192, 32768, 65536, 131072]
for n in sizes:
    sks = [(int.from_bytes(hashlib.sha256(b'sz-sk' + i.to_bytes(4, 'big')).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'sz-m' + i.to_bytes(4, 'big')).digest() for i in range(n)]
    eng.sign_batch_affine(msgs[:64], sks[:64]); eng.get_public_keys(sks[:64])
    eng.sign_batch_affine(msgs, sks)
    ts, tk = [], []
    for _ in range(4):
        t0 = time.perf_counter(); eng.sign_batch_affine(msgs, sks); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); eng.get_public_keys(sks); tk.append(time.perf_counter() - t0)
    eng.timing_enable(True); eng.sign_batch_affine(msgs, sks); tm = eng.timing_read(); eng.timing_enable(False)
    ksum = sum(v[0] for v in tm.values())
    # the C-ABI call by itself (inputs packed, outputs allocated beforehand) and the device-resident entry point
    import ctypes as C
    import numpy as np
    import torch
    blob = b''.join(msgs); kb = b''.join(sks)
    offs = np.zeros(n + 1, dtype=np.uint32); offs[1:] = np.cumsum([len(m) for m in msgs])
    co = (C.c_uint32 * (n + 1)).from_buffer(offs); out = C.create_string_buffer(192 * n); st = C.create_string_buffer(n)
    tp, tr = [], []
    d_m = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda(); d_o = torch.from_numpy(offs.view(np.int32)).cuda(); d_k = torch.frombuffer(bytearray(kb), dtype=torch.uint8).cuda()
    d_so = torch.empty(192 * n, dtype=torch.uint8, device='cuda'); d_ss = torch.empty(n, dtype=torch.uint8, device='cuda')
    for i in range(5):
        t0 = time.perf_counter(); eng.sign_packed(n, blob, co, kb, out, st); tp.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); eng.sign_batch_dev(n, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), d_so.data_ptr(), d_ss.data_ptr()); tr.append(time.perf_counter() - t0)
    print('n=%6d  sign %.3f ms (%.2f M sigs/s) | packed %.3f ms (%.2f M) | resident %.3f ms (%.2f M) | kernels %.3f ms  | getPublicKey %.3f ms (%.2f M keys/s)' % (
        n, min(ts) * 1e3, n / min(ts) / 1e6, min(tp[1:]) * 1e3, n / min(tp[1:]) / 1e6, min(tr[1:]) * 1e3, n / min(tr[1:]) / 1e6, ksum, min(tk) * 1e3, n / min(tk) / 1e6), flush=True)
