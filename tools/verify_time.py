#!/usr/bin/env python3
"""Wall time of verifyBatch(n signatures) with inputs resident in HBM (expand_message_xmd on the device, as in bench.py's verify leg): best and median of k calls, one at a time.  For A/B runs of the decode / hash chain.  Usage: tools/verify_time.py [n] [k]"""
import hashlib, importlib, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = pkg.Engine(0)
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
sks = [(int.from_bytes(hashlib.sha256(b'k' + i.to_bytes(4, 'big')).digest(), 'big') % (R - 1) + 1).to_bytes(32, 'big') for i in range(n)]
msgs = [hashlib.sha256(b'm' + i.to_bytes(4, 'big')).digest() for i in range(n)]
pks = eng.get_public_keys(sks)
aff, st = eng.sign_batch_affine(msgs, sks)
agg, z = eng.point_sum(aff, g2=True); sig = eng.compress_g2(agg)
d_sig = torch.frombuffer(bytearray(sig), dtype=torch.uint8).cuda(); d_pk = torch.frombuffer(bytearray(b''.join(pks)), dtype=torch.uint8).cuda()
d_msgs = torch.frombuffer(bytearray(b''.join(msgs)), dtype=torch.uint8).cuda()
d_off = torch.from_numpy(np.arange(n + 1, dtype=np.uint32) * 32).cuda()
torch.cuda.synchronize()
ts = []
for i in range(k + 2):
    t0 = time.perf_counter(); ok = eng.verify_batch_msgs_dev(n, d_sig.data_ptr(), d_msgs.data_ptr(), d_off.data_ptr(), d_pk.data_ptr()); dt = time.perf_counter() - t0
    assert ok is True
    if i >= 2: ts.append(dt * 1e3)
print('verifyBatch %d: best %.3f ms, median %.3f ms of %d calls' % (n, min(ts), statistics.median(ts), k))
