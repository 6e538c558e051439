"""DESIGN.md quotes numbers from the committed profile files; this keeps the two from drifting apart (CPU only): the rows between the r6-numbers markers must be what
tools/design_numbers.py prints from profiles/round6_bench_*.json, the CPU line likewise, and the per-kernel table of section 4 what tools/design_table.py prints from
profiles/round6_pmc_b*.json.  Round 5 also bounds the document's size (the round-4 review: 92 KB of history; the history now lives in EXPERIMENTS.md)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool)], capture_output=True, text=True, cwd=ROOT, check=True).stdout


def test_section6_rows_match_the_committed_bench_lines():
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    out = _run('design_numbers.py')
    block = design[design.index('<!-- r6-numbers-begin -->'):design.index('<!-- r6-numbers-end -->')]
    rows = [ln for ln in out.splitlines() if ln.startswith('|')]
    assert len(rows) >= 9
    for ln in rows:
        assert ln in block, ln[:80]
    cpu = [ln for ln in out.splitlines() if ln.startswith('CPU:')][0]
    assert cpu in design[design.index('<!-- r6-cpu-begin -->'):design.index('<!-- r6-cpu-end -->')]


def test_section4_kernel_table_matches_the_committed_counters():
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    rows = [ln for ln in _run('design_table.py').splitlines() if ln.startswith('|')]
    assert len(rows) == 12
    for ln in rows:
        assert ln in design, ln[:80]


def test_design_document_stays_a_design_document():
    assert os.path.getsize(os.path.join(ROOT, 'DESIGN.md')) <= 36 * 1024      # (28 KB until round 5; round 6 added four forms -- the wide powers, the wide interpreter, the norm-method square root, the wide G1 sums -- and the soak tests)
