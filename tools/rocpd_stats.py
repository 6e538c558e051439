#!/usr/bin/env python3
"""Summarise a rocprofv3 (--kernel-trace) rocpd sqlite database into a per-kernel stats CSV, grouping the VM kernel's
dispatches by (LDS bytes, grid size) so the step programs can be told apart.  Usage: rocpd_stats.py results.db out.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("""
  select s.kernel_name, d.group_segment_size, d.grid_size_x, d.workgroup_size_x, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
  from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
  group by s.kernel_name, d.group_segment_size, d.grid_size_x order by sum(d.end - d.start) desc""").fetchall()
total = sum(r[5] for r in rows) or 1
with open(sys.argv[2], 'w') as f:
    f.write('kernel,lds_bytes,grid_x,wg_x,calls,total_ns,avg_ns,min_ns,max_ns,pct\n')
    for r in rows:
        f.write('%s,%d,%d,%d,%d,%d,%.0f,%d,%d,%.2f\n' % (r[0].split('(')[0], r[1], r[2], r[3], r[4], r[5], r[5] / r[4], r[6], r[7], 100.0 * r[5] / total))
print(open(sys.argv[2]).read())
