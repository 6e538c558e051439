#!/usr/bin/env python3
"""Rewrites the generated parts of DESIGN.md from the committed profile files: the rows between the r4-numbers / r4-cpu markers (tools/design_numbers.py) and the twelve rows of
the round-4 per-kernel table of section 4 (tools/design_table4.py).  tests/test_docs_numbers.py checks that this has been done.  Usage: tools/refresh_design.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run = lambda t: subprocess.run([sys.executable, os.path.join(ROOT, 'tools', t)], capture_output=True, text=True, cwd=ROOT, check=True).stdout
p = os.path.join(ROOT, 'DESIGN.md')
s = open(p).read()
out = run('design_numbers.py')
table = out[:out.index('\nCPU:')].strip(); cpu = out[out.index('CPU:'):].strip()
a = s.index('<!-- r4-numbers-begin -->') + len('<!-- r4-numbers-begin -->'); b = s.index('<!-- r4-numbers-end -->')
s = s[:a] + '\n| leg | round 4 | fraction of the MAD32 roofline |\n|---|---|---|\n' + table + '\n' + s[b:]
c = s.index('<!-- r4-cpu-begin -->') + len('<!-- r4-cpu-begin -->'); d = s.index('<!-- r4-cpu-end -->')
s = s[:c] + '\n' + cpu + '\n' + s[d:]
rows = run('design_table4.py').strip()
a = s.index('| 4096 | nbls_aot_lines_pq |'); b = s.index('\n\nPer pairing: **173.8 k VALU wave-instructions**')
s = s[:a] + rows + s[b:]
open(p, 'w').write(s)
print('DESIGN.md refreshed')
