"""Randomised parity stress on a GPU box: fully random, pairwise distinct (P, Q) at batch sizes on both sides of every dispatch threshold of
the library (lane-split programs up to 1024, one Miller program below 8192, two halves on two streams from 8192 except 10753..12288, LINES + ACC
from 49152) -- pairings with and without the final exponentiation against the multi-threaded oracle, bilinearity of the Miller product over
the whole batch, MSM against one oracle multiplication.  Usage: python tools/stress_parity.py [size ...]"""
import hashlib, importlib, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import oracle_py
pkg = importlib.import_module('noble-bls12-381_amd')
eng = pkg.Engine(0); o = oracle_py.load(rebuild=False)
g1, g2 = o.g1_generator(), o.g2_generator()
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
bad = 0
SIZES = [int(a) for a in sys.argv[1:]] or [1, 3, 1023, 1024, 1025, 4096, 4097, 8191, 8192, 10752, 10753, 12288, 12289, 20001, 49151, 49152]
for seed, n in enumerate(SIZES):
    rnd = random.Random(2000 + seed)
    ks = [rnd.randrange(1, R) for _ in range(n)]
    # n distinct G1 points through the GPU ladder (already checked against the oracle), n distinct G2 points
    P, st = eng.point_mul_batch([k.to_bytes(32, 'big') for k in ks]); assert not any(st)
    Q, st = eng.point_mul_batch([((k * 7 + 3) % R or 1).to_bytes(32, 'big') for k in ks], pts=g2 * n, g2=True); assert not any(st)
    for fe in (True, False):
        out, _ = eng.pairing_batch(P, Q, fe, False)
        ref, _ = o.pairing_batch(P, Q, fe, False, threads=64)
        if out != ref:
            bad += 1; print('MISMATCH n', n, 'fe', fe)
    # bilinearity over the whole random batch: prod e(k_i G, m_i H) == e(G, H)^(sum k_i m_i)
    t = sum(k * ((k * 7 + 3) % R or 1) for k in ks) % R
    lhs = eng.miller_product(P, Q, True)[0]
    rhs = eng.pairing_batch(o.g1_mul(g1, t)[1], g2, True, False)[0]
    if lhs != rhs: bad += 1; print('BILINEARITY MISMATCH n', n)
    # msm linearity on the same random points
    ms, z = eng.msm(P, [k.to_bytes(32, 'big') for k in ks])
    t2 = sum(k * k for k in ks) % R
    if ms != o.g1_mul(g1, t2)[1]: bad += 1; print('MSM MISMATCH n', n)
    print('n', n, 'checked', flush=True)
print('stress done, sizes', SIZES, 'mismatches', bad)
