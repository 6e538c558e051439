#!/usr/bin/env python3
"""Experiment driver: time the pairing-batch pipeline of an engine library WITHOUT checking results (used with the
NBLS_EXP variants built by tools/exp_variants.sh, some of which compute garbage on purpose to isolate a cost).
Usage: NBLS_LIBRARY=path tools/exp_time.py [batch] [reps]"""
import gzip, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pairs = json.load(gzip.open(os.path.join(ROOT, 'tests', 'golden', 'ref_vectors.json.gz')))['pairs']
g1 = b''.join(bytes.fromhex(v['g1']) for v in pairs); g2 = b''.join(bytes.fromhex(v['g2']) for v in pairs)
m = len(pairs)
G1 = (g1 * (n // m + 1))[:96 * n]; G2 = (g2 * (n // m + 1))[:192 * n]
eng = pkg.Engine(0)
d1 = torch.frombuffer(bytearray(G1), dtype=torch.uint8).cuda(); d2 = torch.frombuffer(bytearray(G2), dtype=torch.uint8).cuda()
out = torch.empty(576 * n, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): eng.pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): eng.pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True, st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
eng.timing_enable(True)
eng.pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True, st); torch.cuda.synchronize()
tm = {k: round(v[0], 4) for k, v in eng.timing_read().items()}
print(os.environ.get('NBLS_LIBRARY', 'default'), 'batch', n, 'ms', round(dt * 1e3, 3), 'pairings/s', round(n / dt), tm)
