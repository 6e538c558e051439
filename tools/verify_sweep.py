#!/usr/bin/env python3
"""Round 5: wall time of verifyBatch(n) (inputs resident in HBM, expand_message_xmd inside) over the settings of the software pipeline
(csrc/pipelines_verify.cpp verify_pipeline): chunks K x size of the last chunk, best and median of k calls each, one call at a time.  NBLS_VERIFY_PIPE=0 in the
environment times round 4's two-phase form with the same script (then the settings are ignored).  Usage: tools/verify_sweep.py [n] [k] [K,K,..] [pct,pct,..]"""
import hashlib, importlib, os, statistics, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '22')   # one hardware queue per stream of the call (the default of 4 would serialise sub-batches that share one)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Ks = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 3, 4, 5, 6]
pcts = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [6, 10, 14, 20]
eng = pkg.Engine(0)
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
sks = [(int.from_bytes(hashlib.sha256(b'k' + i.to_bytes(4, 'big')).digest(), 'big') % (R - 1) + 1).to_bytes(32, 'big') for i in range(n)]
msgs = [hashlib.sha256(b'm' + i.to_bytes(4, 'big')).digest() for i in range(n)]
pks = eng.get_public_keys(sks)
aff, st = eng.sign_batch_affine(msgs, sks)
agg, z = eng.point_sum(aff, g2=True); sig = eng.compress_g2(agg)
d_sig = torch.frombuffer(bytearray(sig), dtype=torch.uint8).cuda(); d_pk = torch.frombuffer(bytearray(b''.join(pks)), dtype=torch.uint8).cuda()
d_msgs = torch.frombuffer(bytearray(b''.join(msgs)), dtype=torch.uint8).cuda()
bad = bytearray(b''.join(msgs)); bad[32 * (n - 1) + 5] ^= 1
d_bad = torch.frombuffer(bad, dtype=torch.uint8).cuda()
d_off = torch.from_numpy(np.arange(n + 1, dtype=np.uint32) * 32).cuda()
torch.cuda.synchronize()


def timed():
    ts = []
    for i in range(k + 2):
        t0 = time.perf_counter(); ok = eng.verify_batch_msgs_dev(n, d_sig.data_ptr(), d_msgs.data_ptr(), d_off.data_ptr(), d_pk.data_ptr()); dt = time.perf_counter() - t0
        assert ok is True
        if i >= 2: ts.append(dt * 1e3)
    assert eng.verify_batch_msgs_dev(n, d_sig.data_ptr(), d_bad.data_ptr(), d_off.data_ptr(), d_pk.data_ptr()) is False
    return min(ts), statistics.median(ts)


if os.environ.get('NBLS_VERIFY_PIPE') == '0':
    print('verifyBatch %d, two-phase form (round 4): best %.3f ms, median %.3f ms' % ((n,) + timed()))
    sys.exit(0)
for K in Ks:
    for pct in (pcts if K > 1 else pcts[:1]):
        eng.set_verify_pipeline(chunks=K, last_pct=pct, pipe_min=0)
        b, m = timed()
        print('verifyBatch %d  K=%d last=%2d%%: best %.3f ms, median %.3f ms' % (n, K, pct, b, m), flush=True)
