#!/usr/bin/env python3
"""Rewrites the generated parts of DESIGN.md from the committed profile files: the rows between the r6-numbers / r6-cpu markers (tools/design_numbers.py) and the twelve rows of
the per-kernel table of section 4 (tools/design_table.py).  tests/test_docs_numbers.py checks that this has been done.  Usage: tools/refresh_design.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run = lambda t: subprocess.run([sys.executable, os.path.join(ROOT, 'tools', t)], capture_output=True, text=True, cwd=ROOT, check=True).stdout
p = os.path.join(ROOT, 'DESIGN.md')
s = open(p).read()
out = run('design_numbers.py')
table = out[:out.index('\nCPU:')].strip(); cpu = out[out.index('CPU:'):].strip()
a = s.index('<!-- r6-numbers-begin -->') + len('<!-- r6-numbers-begin -->'); b = s.index('<!-- r6-numbers-end -->')
s = s[:a] + '\n| leg | round 6 | fraction of the MAD32 roofline |\n|---|---|---|\n' + table + '\n' + s[b:]
c = s.index('<!-- r6-cpu-begin -->') + len('<!-- r6-cpu-begin -->'); d = s.index('<!-- r6-cpu-end -->')
s = s[:c] + '\n' + cpu + '\n' + s[d:]
rows = run('design_table.py').strip()
a = s.index('| 4096 | nbls_aot_lines_pq |'); b = s.index('\n\n**Instructions per pairing**')
s = s[:a] + rows + s[b:]
open(p, 'w').write(s)
print('DESIGN.md refreshed')
