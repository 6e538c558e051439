#!/usr/bin/env python3
"""Wall time of one Miller product over n device-resident pairs with one final exponentiation (nbls_miller_product_dev; BASELINE configs[4] on one GPU): k calls, one at a time.
The pairs cycle through (P_a, Q_b) and (-P_a, Q_b), so the product must be ONE.  Usage: tools/product_time.py [n] [k]"""
import gzip, importlib, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
import oracle_py
oracle = oracle_py.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pairs = json.load(gzip.open(os.path.join(ROOT, 'tests', 'golden', 'ref_vectors.json.gz')))['pairs']
g1 = [bytes.fromhex(v['g1']) for v in pairs]; g2 = [bytes.fromhex(v['g2']) for v in pairs]
neg = [oracle.un('g1_neg_aff', p, 96) for p in g1]
m = len(pairs)
half = n // 2
G1 = b''.join(g1[i % m] for i in range(half)) + b''.join(neg[i % m] for i in range(half))
G2 = b''.join(g2[(3 * i + 1) % m] for i in range(half)) * 2
eng = pkg.Engine(0)
d1 = torch.frombuffer(bytearray(G1), dtype=torch.uint8).cuda(); d2 = torch.frombuffer(bytearray(G2), dtype=torch.uint8).cuda()
out = torch.empty(576, dtype=torch.uint8, device='cuda')
ts = []
for i in range(k + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.miller_product_dev(2 * half, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True); torch.cuda.synchronize()
    if i: ts.append((time.perf_counter() - t0) * 1e3)
assert bytes(out.cpu().numpy().tobytes()) == bytes(47) + b'\x01' + bytes(528), 'product is not ONE'
print('miller_product %d pairs: best %.3f ms, median %.3f ms of %d calls (%.2f M terms/s)' % (2 * half, min(ts), statistics.median(ts), k, 2 * half / min(ts) / 1e3))
