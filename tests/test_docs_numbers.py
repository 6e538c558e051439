"""DESIGN.md quotes numbers from the committed profile files; this keeps the two from drifting apart (CPU only): the rows between the r4-numbers markers must be what
tools/design_numbers.py prints from profiles/round4_bench_*.json, the CPU line likewise, and the per-kernel table of section 4 what tools/design_table4.py prints from
profiles/round4_pmc_b*.json."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool)], capture_output=True, text=True, cwd=ROOT, check=True).stdout


def test_section6_rows_match_the_committed_bench_lines():
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    out = _run('design_numbers.py')
    block = design[design.index('<!-- r4-numbers-begin -->'):design.index('<!-- r4-numbers-end -->')]
    rows = [ln for ln in out.splitlines() if ln.startswith('|')]
    assert len(rows) >= 8
    for ln in rows:
        assert ln in block, ln[:80]
    cpu = [ln for ln in out.splitlines() if ln.startswith('CPU:')][0]
    assert cpu in design[design.index('<!-- r4-cpu-begin -->'):design.index('<!-- r4-cpu-end -->')]


def test_section4_kernel_table_matches_the_committed_counters():
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    rows = [ln for ln in _run('design_table4.py').splitlines() if ln.startswith('|')]
    assert len(rows) == 12
    for ln in rows:
        assert ln in design, ln[:80]
