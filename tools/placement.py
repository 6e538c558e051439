#!/usr/bin/env python3
"""Where does the dispatcher put the single-wavefront workgroups of a launch?  Histogram of workgroups per SIMD / per CU for
n work items (4 per workgroup).  Usage: [NBLS_LDS_FLOOR=bytes] tools/placement.py n"""
import collections, ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('noble-bls12-381_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = pkg.Engine(0)
blocks = (n + 3) // 4
out5 = (C.c_uint64 * (5 * blocks))()
eng.lib.nbls_placement_probe.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
r = eng.lib.nbls_placement_probe(eng.h, n, out5)
assert r == 0, r
out = [v for b in range(blocks) for v in out5[5 * b:5 * b + 3]]      # (hw id, start tick, end tick) per workgroup
real = [(out5[5 * b + 3], out5[5 * b + 4]) for b in range(blocks)]     # s_memrealtime (100 MHz) at start / end
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
# co-residency: for SIMDs that received two workgroups, how much of their lifetimes overlapped (same XCD -> same counter)
by_simd = collections.defaultdict(list)
for b in range(blocks):
    v = out[3 * b]; hw, xcc = v & 0xffffffff, (v >> 32) & 0xf
    by_simd[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3)].append((out[3 * b + 1], out[3 * b + 2]))
ov = []
for iv in by_simd.values():
    if len(iv) == 2:
        (s0, e0), (s1, e1) = iv
        ov.append(max(0, min(e0, e1) - max(s0, s1)) / max(e0 - s0, e1 - s1))
if ov:
    ov.sort(); print('  two-workgroup SIMDs: overlap fraction p10 %.2f p50 %.2f p90 %.2f' % (ov[len(ov) // 10], ov[len(ov) // 2], ov[9 * len(ov) // 10]))
    ds = sorted(abs(iv[0][0] - iv[1][0]) for iv in by_simd.values() if len(iv) == 2)
    print('  start-time difference of the two (ticks): p10 %d p50 %d p90 %d' % (ds[len(ds) // 10], ds[len(ds) // 2], ds[9 * len(ds) // 10]))
# launch timeline inside XCD 0: start offsets (deciles) relative to its first wave, in units of the median wave lifetime
x0 = sorted(out[3 * b + 1] for b in range(blocks) if ((out[3 * b] >> 32) & 0xf) == 0)
if x0:
    med = sorted(out[3 * b + 2] - out[3 * b + 1] for b in range(blocks))[blocks // 2]
    print('  XCD 0: %d workgroups, start offsets / median life at deciles: %s' % (len(x0), ' '.join('%.2f' % ((x0[min(len(x0) - 1, len(x0) * d // 10)] - x0[0]) / med) for d in range(11))))
q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]
print('  ticks: start p50 %d p90 %d max %d | wave life p50 %d max %d | last end %d' % (q(starts, .5), q(starts, .9), starts[-1], q(lives, .5), lives[-1], ends[-1]))

# real time (100 MHz counter, common to the whole device): when do the workgroups start and end, and how fast does s_memtime tick meanwhile
r0 = min(a for a, _ in real)
st = sorted((a - r0) / 100.0 for a, _ in real); en = sorted((e - r0) / 100.0 for _, e in real); lf = sorted((e - a) / 100.0 for a, e in real)
print('  real time (us): start p10 %.1f p50 %.1f p90 %.1f max %.1f | end p10 %.1f p50 %.1f max %.1f | life p10 %.1f p50 %.1f p90 %.1f' % (q(st, .1), q(st, .5), q(st, .9), st[-1], q(en, .1), q(en, .5), en[-1], q(lf, .1), q(lf, .5), q(lf, .9)))
rates = sorted((out5[5 * b + 2] - out5[5 * b + 1]) / max(1, (out5[5 * b + 4] - out5[5 * b + 3])) * 100.0 for b in range(blocks))
print('  s_memtime ticks per microsecond of real time: p10 %.0f p50 %.0f p90 %.0f' % (q(rates, .1), q(rates, .5), q(rates, .9)))
