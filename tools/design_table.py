#!/usr/bin/env python3
"""Per-program table for DESIGN.md section 4, generated from the committed profiles: VALU instructions per wavefront and kernel time from the PMC summaries
(profiles/round3_pmc_b*.csv), issue-slot use, multiply-add share from the host compiler's product / reduction counts, registers from the code-object metadata
(profiles/round3_kernel_resources.txt), and the roofline fraction of each program = algorithmic MAD32 of the reference routine it implements / time / peak.
Usage: tools/design_table.py > profiles/round3_program_table.md"""
import csv, ctypes, io, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 256 * 64 * 2.4e9
# algorithmic Fp multiplications of the reference routine behind each program; the final exponentiation's 12,166 are split by the host compiler's
# product counts of its phase programs
ALG = {'miller_fe': 7556, 'lines_pq': 2400, 'acc_fe': 5156}
res = {}
for ln in open(os.path.join(ROOT, 'profiles', 'round3_kernel_resources.txt')):
    m = re.match(r'\S+\s+(\S+)\s+vgpr (\d+)', ln)
    if m: res[m.group(1)] = int(m.group(2))
# host compiler statistics (products, lane-ops, items per wavefront) through the simulator library
out = subprocess.run([sys.executable, '-c', "import ctypes; l=ctypes.CDLL('%s'); l.nbls_sim_stats()" % os.path.join(ROOT, 'noble-bls12-381_amd', 'libnbls_sim.so')], capture_output=True, text=True).stdout
st = {}
for ln in out.splitlines():
    m = re.match(r'(\S+)\s+W=\s*(\d+) G=(\d+) steps=\s*(\d+) \(dot\s+(\d+),.*dot_ops=\s*(\d+) products=\s*(\d+).*slots=\s*(\d+)\s+lds=\s*(\d+).*round operands (\d+):', ln)
    if m: st[m.group(1)] = dict(W=int(m.group(2)), G=int(m.group(3)), steps=int(m.group(4)), dot_steps=int(m.group(5)), dot_ops=int(m.group(6)), products=int(m.group(7)), slots=int(m.group(8)), lds=int(m.group(9)), rounds=int(m.group(10)) // 2)
fe_prog = ['fe_easy', 'expx', 'fe_mid1', 'fe_mid2', 'fe_final']
fe_mult = {'expx': 5}
fe_products = sum((st[p]['products'] + st[p]['dot_ops']) * fe_mult.get(p, 1) for p in fe_prog)
for p in fe_prog: ALG[p] = 12166 * (st[p]['products'] + st[p]['dot_ops']) / fe_products
print('| batch | program | lanes × items per wave | VALU instr per wave | multiply-adds among them | waves | LDS per wave | kernel µs (alone) | issue slots used | algorithmic Fp-mul per item | roofline frac |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
for n, path in ((4096, 'round3_pmc_b4096.csv'), (65536, 'round3_pmc_b65536.csv')):
    for r in csv.DictReader(open(os.path.join(ROOT, 'profiles', path))):
        p = r['program']
        if p == 'fp_inv':
            print('| %d | fp_inv (`nbls_fp_inv_kernel`, %d VGPRs) | 1 × 64 | %d | – | %d | 0 | %.1f | %.2f | – | – |' % (n, res.get('nbls_fp_inv_kernel', 0), float(r['SQ_INSTS_VALU']) / float(r['SQ_WAVES']), float(r['SQ_WAVES']), float(r['avg_us_under_pmc']), float(r['SQ_INSTS_VALU']) * 4 / (float(r['avg_us_under_pmc']) * 1e-6 * 2.4e9 * 1024)))
            continue
        s = st[p]
        valu = float(r['SQ_INSTS_VALU']) / float(r['SQ_WAVES'])
        mad = 196 * (s['rounds'] + s['dot_steps'])   # multiply-add wave-instructions: 196 per product round and per reduction of every DOT step (a round is as long as its heaviest lane)
        us = float(r['avg_us_under_pmc'])
        busy = float(r['SQ_INSTS_VALU']) * 4 / (us * 1e-6 * 2.4e9 * 1024)
        frac = n * ALG[p] * 300 / (us * 1e-6) / PEAK
        print('| %d | %s | %d × %d | %d | %s | %d | %d B | %.1f | %.2f | %.0f | %.3f |' % (n, p, s['W'], s['G'], valu, '%d %%' % round(100 * mad / valu), float(r['SQ_WAVES']), s['lds'], us, busy, ALG[p], frac))
