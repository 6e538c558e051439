#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the per-program PMC summaries (tools/pmc_summary.py output, one CSV per batch size).

Per batch size: HBM bytes of one pairing call = sum over its launches of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE / WRITE_SIZE are in KB,
FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md prescribes -- with EXPX counted five times; the VALU wave-instructions of the call and
valu_issue_busy = SQ_INSTS_VALU x 4 clocks / (kernel time under the counter pass x 2.4 GHz x 1024 SIMDs); per program: VALU instructions per wavefront,
waves, mean duration, share of LDS cycles lost to bank conflicts.  bench.py reads bytes_per_step and valu_issue_busy for its roofline object.
Usage: tools/hbm_traffic.py profiles/round2_pmc_b4096.csv profiles/round2_pmc_b65536.csv > profiles/hbm_traffic.json"""
import csv, json, sys
ALG = lambda n: n * (288 + 832 + 128 + 832 + 768 + 5 * 1536 + 2 * 2304 + 5376 + 576 + (2 * 26112 if n >= 49152 else 0))   # bench.py, DESIGN.md section 4
out = {}
for path in sys.argv[1:]:
    rows = {r['program']: r for r in csv.DictReader(open(path))}
    n = 65536 if 'lines_pq' in rows else 4096
    mult = lambda p: 5 if p == 'expx' else 1
    f = lambda r, k: float(r.get(k) or 0)
    fetch = sum(f(r, 'FETCH_SIZE') * 1024 * mult(p) for p, r in rows.items())
    write = sum(f(r, 'WRITE_SIZE') * 1024 * mult(p) for p, r in rows.items())
    valu = sum(f(r, 'SQ_INSTS_VALU') * mult(p) for p, r in rows.items())
    t_us = sum(f(r, 'avg_us_under_pmc') * mult(p) for p, r in rows.items())
    progs = {}
    for p, r in rows.items():
        waves = f(r, 'SQ_WAVES') or 1
        progs[p] = {'valu_per_wave': round(f(r, 'SQ_INSTS_VALU') / waves), 'waves': round(waves), 'avg_us_under_pmc': f(r, 'avg_us_under_pmc'),
                    'lds_conflict_frac': round(f(r, 'SQ_LDS_BANK_CONFLICT') / f(r, 'SQ_LDS_IDX_ACTIVE'), 3) if f(r, 'SQ_LDS_IDX_ACTIVE') else 0.0}
    out[str(n)] = {'bytes_per_step': round(2 * fetch + write), 'fetch_size_bytes_raw': round(fetch), 'write_size_bytes': round(write),
                   'valu_wave_instructions_per_step': round(valu), 'valu_issue_busy': round(valu * 4 / (t_us * 1e-6 * 2.4e9 * 1024), 4),
                   'algorithmic_bytes_per_step': ALG(n), 'programs': progs,
                   'note': 'round 3 (tools/profile_round3.sh -> tools/pmc_summary.py -> tools/hbm_traffic.py): sums over the launches of one pairing call (expx x5); FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; valu_issue_busy = SQ_INSTS_VALU x 4 clocks / (kernel time under the counter pass x 2.4 GHz x 1024 SIMDs)'}
json.dump(out, sys.stdout, indent=1)
