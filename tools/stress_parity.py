"""Randomised parity stress on a GPU box: several seeds of 4096 fully random, pairwise distinct (P, Q) -- pairings with and without
the final exponentiation against the multi-threaded oracle, bilinearity of the Miller product over the whole batch, MSM against one
oracle multiplication.  Usage: python tools/stress_parity.py"""
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
for seed in range(4):
    rnd = random.Random(1000 + seed)
    n = 4096
    ks = [rnd.randrange(1, R) for _ in range(n)]
    # n distinct G1 points through the GPU ladder (already checked against the oracle), n distinct G2 points
    P, st = eng.point_mul_batch([k.to_bytes(32, 'big') for k in ks]); assert not any(st)
    Q, st = eng.point_mul_batch([((k * 7 + 3) % R or 1).to_bytes(32, 'big') for k in ks], pts=g2 * n, g2=True); assert not any(st)
    for fe in (True, False):
        out, _ = eng.pairing_batch(P, Q, fe, False)
        ref, _ = o.pairing_batch(P, Q, fe, False, threads=64)
        if out != ref:
            bad += 1; print('MISMATCH seed', seed, 'fe', fe)
    # bilinearity over the whole random batch: prod e(k_i G, m_i H) == e(G, H)^(sum k_i m_i)
    t = sum(k * ((k * 7 + 3) % R or 1) for k in ks) % R
    lhs = eng.miller_product(P, Q, True)[0]
    rhs = eng.pairing_batch(o.g1_mul(g1, t)[1], g2, True, False)[0]
    if lhs != rhs: bad += 1; print('BILINEARITY MISMATCH seed', seed)
    # msm linearity on the same random points
    ms, z = eng.msm(P, [k.to_bytes(32, 'big') for k in ks])
    t2 = sum(k * k for k in ks) % R
    if ms != o.g1_mul(g1, t2)[1]: bad += 1; print('MSM MISMATCH seed', seed)
print('stress done, mismatches', bad)
