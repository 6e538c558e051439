#!/usr/bin/env python3
"""Per-kernel time breakdown of verifyBatch(n signatures) with device-resident inputs. Usage: tools/verify_breakdown.py [n]"""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = pkg.Engine(0)
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
sks = [(int.from_bytes(hashlib.sha256(b'k' + i.to_bytes(4, 'big')).digest(), 'big') % (R - 1) + 1).to_bytes(32, 'big') for i in range(n)]
msgs = [hashlib.sha256(b'm' + i.to_bytes(4, 'big')).digest() for i in range(n)]
t0 = time.perf_counter(); pks = eng.get_public_keys(sks); t1 = time.perf_counter()
aff, st = eng.sign_batch_affine(msgs, sks); t2 = time.perf_counter()
agg, z = eng.point_sum(aff, g2=True); sig = eng.compress_g2(agg)
print('setup on GPU: keys %.1f ms, signatures %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
assert eng.verify_batch(sig, msgs, pks) is True
import oracle_py
oracle = oracle_py.load()
uni = b''.join(oracle.expand_message_xmd(m, oracle_py.DST_DEFAULT, 256) for m in msgs)
d_sig = torch.frombuffer(bytearray(sig), dtype=torch.uint8).cuda(); d_uni = torch.frombuffer(bytearray(uni), dtype=torch.uint8).cuda(); d_pk = torch.frombuffer(bytearray(b''.join(pks)), dtype=torch.uint8).cuda()
assert eng.verify_batch_dev(n, d_sig.data_ptr(), d_uni.data_ptr(), d_pk.data_ptr()) is True
if os.environ.get('NBLS_VB_NOTIMING'):     # for a kernel trace of the call as it runs in production (per-kernel HIP events make the host wait between launches)
    torch.cuda.synchronize(); time.sleep(0.05)
    t0 = time.perf_counter(); eng.verify_batch_dev(n, d_sig.data_ptr(), d_uni.data_ptr(), d_pk.data_ptr()); dt = time.perf_counter() - t0
    print('verify_batch_dev %d: %.2f ms (no per-kernel timing)' % (n, dt * 1e3)); sys.exit(0)
eng.timing_enable(True)
t0 = time.perf_counter(); eng.verify_batch_dev(n, d_sig.data_ptr(), d_uni.data_ptr(), d_pk.data_ptr()); dt = time.perf_counter() - t0
tm = eng.timing_read()
print('verify_batch_dev %d: %.2f ms' % (n, dt * 1e3))
for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0]): print('  %-14s %8.3f ms  x%d' % (k, v[0], v[1]))
print('  sum kernels %.2f ms' % sum(v[0] for v in tm.values()))
