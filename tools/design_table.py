#!/usr/bin/env python3
"""Rows of the per-kernel table of DESIGN.md section 4 from the committed PMC summaries (profiles/roundN_pmc_b4096.json / _b65536.json, tools/pmc_percall.py; N = argv[1], default 6).
Static columns: lanes x items per wavefront and the multiply-add share (196 x (product rounds + reductions) of the program's K_DOT steps / measured VALU instructions --
instruction counts do not change from box to box); algorithmic Fp multiplications per item as in SURVEY 8(d).  Usage: tools/design_table4.py"""
import json, os, sys
RND = sys.argv[1] if len(sys.argv) > 1 else '6'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 256 * 64 * 2.4e9
K = [('nbls_aot_lines_pq', '10 × 6', '63 %', 2400), ('nbls_aot_acc_fe', '12 × 5', '79 %', 5156), ('nbls_fp_inv_kernel', '1 × 64', '–', None), ('nbls_aot_fe_easy', '16 × 4', '77 %', 374),
     ('nbls_aot_expx', '12 × 5', '73 %', 11025), ('nbls_aot_fe_final', '32 × 2', '74 %', 767)]
for b in (4096, 65536):
    j = json.load(open(os.path.join(ROOT, 'profiles', 'round%s_pmc_b%d.json' % (RND, b))))['kernels']
    for name, shape, share, alg in K:
        if name == 'nbls_fp_inv_kernel' and name not in j: name, shape = 'nbls_fp_inv_wide_kernel', '16 × 4'      # launches of at most 4096 elements: one limb per lane (fp_inv_wide.h)
        v = j[name]; L = v['launches_per_call']; us = v['avg_us_under_pmc'] * L; valu = v['valu_per_wave'] * L
        label = name + (' (chain)' if name == 'nbls_aot_expx' and L == 1 else ' (%d launches)' % L if L > 1 else '')
        frac = '%.3f' % (alg * 300 * b / (us * 1e-6) / PEAK) if alg else '–'
        lds = '%.2f' % v['lds_conflict_frac'] if v.get('lds_conflict_frac') is not None else '–'
        print('| %d | %s | %s | %d | %s | %d | %.1f | %.2f | %s | %s | %s |' % (b, label, shape, valu, share, v['waves'], us, v['issue_slots_used'], lds, alg if alg else '–', frac))
