#!/usr/bin/env python3
"""Static ISA statistics of a gfx950 assembly listing (hipcc -S): per kernel, per basic block with >= MIN VALU instructions, the
counts of VALU / multiply-add / LDS / VMEM / SALU / waitcnt instructions.  Used to compare a specialised step body of the ahead-of-time
kernels (aot_kernel.hip) with the interpreter's (vm_kernel.hip).  usage: isa_blocks.py file.s [kernel-substring] [min_valu]"""
import re, sys
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ''; minv = int(sys.argv[3]) if len(sys.argv) > 3 else 100
kernel = None; blocks = []; cur = None
def flush():
    global cur
    if cur and cur['n']: blocks.append(cur)
    cur = None
for line in open(path):
    s = line.strip()
    m = re.match(r'^([A-Za-z_.][\w.$]*):', line)
    if m and not line.startswith('\t'):
        name = m.group(1)
        if not name.startswith('.L'):
            flush(); kernel = name
        flush(); cur = dict(k=kernel, label=name, n=0, valu=0, madi=0, madu=0, ds=0, vmem=0, salu=0, wait=0, br=0, mov=0, nop=0)
        continue
    if cur is None or not s or s.startswith(('.', ';')): continue
    op = s.split()[0]
    if not re.match(r'^[a-z]', op): continue
    cur['n'] += 1
    if op.startswith('v_'):
        cur['valu'] += 1
        if op == 'v_mad_i64_i32': cur['madi'] += 1
        elif op == 'v_mad_u64_u32': cur['madu'] += 1
        elif op.startswith('v_mov') or op.startswith('v_accvgpr'): cur['mov'] += 1
    elif op.startswith('ds_'): cur['ds'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur['vmem'] += 1
    elif op.startswith('s_waitcnt'): cur['wait'] += 1
    elif op.startswith('s_nop'): cur['nop'] += 1
    elif op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_swappc')): cur['br'] += 1; cur['salu'] += 1
    elif op.startswith('s_'): cur['salu'] += 1
flush()
tot = {}
for b in blocks:
    if want and want not in (b['k'] or ''): continue
    t = tot.setdefault(b['k'], dict(n=0, valu=0, madi=0, madu=0))
    for k in t: t[k] += b[k]
    if b['valu'] >= minv:
        print('%-22s %-12s instr %5d  VALU %5d  mad_i64 %4d  mad_u64 %4d  other-VALU %4d (mov %3d)  ds %3d  vmem %2d  salu %3d  wait %2d  nop %2d' % (b['k'], b['label'], b['n'], b['valu'], b['madi'], b['madu'], b['valu'] - b['madi'] - b['madu'], b['mov'], b['ds'], b['vmem'], b['salu'], b['wait'], b['nop']))
for k, t in tot.items(): print('== %s: %d instructions (%d bytes approx), VALU %d, multiply-adds %d' % (k, t['n'], 8 * t['n'], t['valu'], t['madi'] + t['madu']))
