import hashlib, importlib, os, sys, time
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('noble-bls12-381_amd')
eng = pkg.Engine(0)
n=8192
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
sks = [(int.from_bytes(hashlib.sha256(b'k' + i.to_bytes(4, 'big')).digest(), 'big') % (R - 1) + 1).to_bytes(32, 'big') for i in range(n)]
msgs = [hashlib.sha256(b'm' + i.to_bytes(4, 'big')).digest() for i in range(n)]
eng.point_mul_batch(sks); eng.sign_batch_affine(msgs, sks)
eng.timing_enable(True)
t0=time.perf_counter(); eng.point_mul_batch(sks); t1=time.perf_counter(); eng.sign_batch_affine(msgs, sks); t2=time.perf_counter()
tm=eng.timing_read()
print('getpk %.2f ms sign %.2f ms'%((t1-t0)*1e3,(t2-t1)*1e3), {k:round(v[0],2) for k,v in tm.items() if v[0]>0.3})
