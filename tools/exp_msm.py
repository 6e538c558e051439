#!/usr/bin/env python3
"""Time the device-resident multi-scalar multiplication (nbls_msm_dev) and print the per-kernel breakdown.
Usage: tools/exp_msm.py [n] [g2: 0|1] [nbits] [reps]"""
import gzip, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g2 = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
nbits = int(sys.argv[3]) if len(sys.argv) > 3 else 255
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
pairs = json.load(gzip.open(os.path.join(ROOT, 'tests', 'golden', 'ref_vectors.json.gz')))['pairs']
key, sz = ('g2', 192) if g2 else ('g1', 96)
pts = b''.join(bytes.fromhex(v[key]) for v in pairs)
P = (pts * (n // len(pairs) + 1))[:sz * n]
eng = pkg.Engine(0)
d_p = torch.frombuffer(bytearray(P), dtype=torch.uint8).cuda()
g = torch.Generator(device='cpu'); g.manual_seed(1)
k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
drop = (256 - nbits) // 8
if drop: k[:, :drop] = 0
if (256 - nbits) % 8: k[:, drop] &= (1 << (8 - (256 - nbits) % 8)) - 1
d_k = k.cuda()
out = torch.empty(sz, dtype=torch.uint8, device='cuda'); st = torch.empty(1, dtype=torch.int8, device='cuda')
s = torch.cuda.current_stream().cuda_stream
for _ in range(2): eng.msm_dev(g2, n, d_p.data_ptr(), d_k.data_ptr(), nbits, out.data_ptr(), st.data_ptr(), s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): eng.msm_dev(g2, n, d_p.data_ptr(), d_k.data_ptr(), nbits, out.data_ptr(), st.data_ptr(), s)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
eng.timing_enable(True)
eng.msm_dev(g2, n, d_p.data_ptr(), d_k.data_ptr(), nbits, out.data_ptr(), st.data_ptr(), s); torch.cuda.synchronize()
tm = {k: (round(v[0], 3), v[1]) for k, v in eng.timing_read().items() if v[1]}
print('msm', 'G2' if g2 else 'G1', 'n', n, 'nbits', nbits, 'ms', round(dt * 1e3, 3), 'points/s', round(n / dt), 'vm kernels (ms, launches):', tm)
