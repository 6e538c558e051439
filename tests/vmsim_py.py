"""ctypes binding of noble-bls12-381_amd/libnbls_sim.so: host simulator of the wave VM (test infrastructure)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'noble-bls12-381_amd')
PROGS = ['MILLER_BYTES', 'MILLER_RAW', 'MILLER_FE', 'NORM_RAW', 'NORM_BYTES', 'FE_EASY', 'EXPX', 'FE_MID1', 'FE_MID2', 'FE_FINAL', 'MUL2', 'RAW_TO_BYTES']
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


def final_exp(lib, n, F, N, out):
    """The launch sequence of final_exp_pipeline() in csrc/nbls_api.cpp, on the simulator."""
    NI = C.create_string_buffer(48 * n)
    T = [C.create_string_buffer(576 * n) for _ in range(7)]
    lib.nbls_sim_fp_inv(C.c_uint(n), N, NI)
    run(lib, 'FE_EASY', n, {3: (F, 576), 4: (NI, 48), 5: (T[0], 576)})
    run(lib, 'EXPX', n, {3: (T[0], 576), 5: (T[1], 576)})
    run(lib, 'FE_MID1', n, {3: (T[0], 576), 5: (T[1], 576), 6: (T[2], 576)})
    run(lib, 'EXPX', n, {3: (T[2], 576), 5: (T[3], 576)})
    run(lib, 'EXPX', n, {3: (T[3], 576), 5: (T[4], 576)})
    run(lib, 'EXPX', n, {3: (T[4], 576), 5: (T[6], 576)})
    run(lib, 'FE_MID2', n, {3: (T[6], 576), 5: (T[1], 576), 6: (T[5], 576)})
    run(lib, 'EXPX', n, {3: (T[5], 576), 5: (T[6], 576)})
    bufs = {i: (T[i], 576) for i in range(7)}
    bufs[7] = (out, 576)
    run(lib, 'FE_FINAL', n, bufs)
