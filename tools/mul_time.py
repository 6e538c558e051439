import hashlib, importlib, os, sys, time
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module('noble-bls12-381_amd')
eng = pkg.Engine(0)
for n in (8192, 65536):
    sks = [(int.from_bytes(hashlib.sha256(b'k' + i.to_bytes(4, 'big')).digest(), 'big') % (2**254) + 1).to_bytes(32, 'big') for i in range(n)]
    msgs = [hashlib.sha256(b'm' + i.to_bytes(4, 'big')).digest() for i in range(n)]
    eng.get_public_keys(sks[:64]); eng.sign_batch_affine(msgs[:64], sks[:64])
    t0 = time.perf_counter(); eng.get_public_keys(sks); t1 = time.perf_counter(); eng.sign_batch_affine(msgs, sks); t2 = time.perf_counter()
    eng.timing_enable(True); eng.get_public_keys(sks); a = eng.timing_read(); eng.sign_batch_affine(msgs, sks); b = eng.timing_read(); eng.timing_enable(False)
    print(n, 'getPublicKey %.2f ms (%.2f M/s)' % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6), 'sign %.2f ms (%.2f M/s)' % ((t2 - t1) * 1e3, n / (t2 - t1) / 1e6))
    print('  gpk kernels', {k: round(v[0], 3) for k, v in a.items()}); print('  sign kernels', {k: round(v[0], 3) for k, v in b.items()})
