"""ctypes binding of noble-bls12-381_amd/libnbls_sim.so: host simulator of the wave VM (test infrastructure)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'noble-bls12-381_amd')
PROGS = ['MILLER_BYTES', 'MILLER_RAW', 'MILLER_FE', 'NORM_RAW', 'NORM_BYTES', 'FE_HARD', 'MUL2', 'RAW_TO_BYTES']
P = {n: i for i, n in enumerate(PROGS)}


def load():
    subprocess.check_call(['make', '-s', '-C', os.path.join(PKG, 'csrc'), '../libnbls_sim.so'])
    return C.CDLL(os.path.join(PKG, 'libnbls_sim.so'))


def run(lib, prog, n, bufs):
    """bufs: dict index -> (ctypes buffer, stride)"""
    ptrs = (C.c_void_p * 8)()
    strides = (C.c_uint64 * 8)()
    for k, (b, s) in bufs.items():
        ptrs[k] = C.cast(b, C.c_void_p)
        strides[k] = s
    r = lib.nbls_sim_run(P[prog], C.c_uint(n), ptrs, strides)
    assert r == 0
