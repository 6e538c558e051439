#!/usr/bin/env python3
"""Where does the dispatcher put the single-wavefront workgroups of a launch?  Histogram of workgroups per SIMD / per CU for
n work items (4 per workgroup).  Usage: [NBLS_LDS_FLOOR=bytes] [NBLS_SPLIT=0] tools/placement.py n"""
import collections, ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('noble-bls12-381_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = pkg.Engine(0)
blocks = (n + 3) // 4
out = (C.c_uint64 * (3 * blocks))()
eng.lib.nbls_placement_probe.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
r = eng.lib.nbls_placement_probe(eng.h, n, out)
assert r == 0, r
simd = collections.Counter(); cu = collections.Counter()
t0 = min(out[3 * b + 1] for b in range(blocks))
starts = sorted(out[3 * b + 1] - t0 for b in range(blocks)); ends = sorted(out[3 * b + 2] - t0 for b in range(blocks)); lives = sorted(out[3 * b + 2] - out[3 * b + 1] for b in range(blocks))
for v in out[0::3]:
    hw, xcc = v & 0xffffffff, (v >> 32) & 0xf
    wave, sd, cuid, sh, se = hw & 0xf, (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    simd[(xcc, se, sh, cuid, sd)] += 1; cu[(xcc, se, sh, cuid)] += 1
print('n', n, 'workgroups', blocks, 'lds floor', os.environ.get('NBLS_LDS_FLOOR', '-'))
print('  distinct CUs', len(cu), 'distinct SIMDs', len(simd))
print('  workgroups per SIMD histogram', sorted(collections.Counter(simd.values()).items()))
print('  workgroups per CU histogram  ', sorted(collections.Counter(cu.values()).items()))
q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]
print('  ticks: start p50 %d p90 %d max %d | wave life p50 %d max %d | last end %d' % (q(starts, .5), q(starts, .9), starts[-1], q(lives, .5), lives[-1], ends[-1]))
