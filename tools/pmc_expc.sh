#!/bin/bash
# Counters of the compressed-squaring program against the plain exponentiation program in one run (65,536 pairings, kernels alone): why does EXPC_SQ retire
# more instructions per millisecond than EXPX?   Output: gpurun_out/pmc_expc/summary.txt
export TMPDIR=/tmp NBLS_HALVES_MIN=0
out=$PWD/gpurun_out/pmc_expc; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  NBLS_EXPC_MIN=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- python tools/exp_time.py 65536 1 > $out/run$i.log 2>&1
done
python - $out <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pmc*/**/*_counter_collection.csv', recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('nbls_vm_kernel')]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    grid = {int(r['Dispatch_Id']): int(r['Grid_Size']) for r in rows}
    # exp_time.py runs the call four times (2 warm-up, 1 timed, 1 with timing); per call: lines (n/6 waves), acc, fe_easy, then per exponentiation sq (n/8 waves), dec_a, dec_b, expx over the redo list
    label = {}
    for d in ids:
        w = grid[d] // 64
        label[d] = 'expc_sq' if w == 8192 else None
    seen = set()
    prev = None
    for d in ids:
        if label[d] is None:
            # the three launches after an expc_sq are dec_a, dec_b, expx(redo)
            k = [x for x in ids if x < d and label.get(x) == 'expc_sq']
            if k:
                pos = len([x for x in ids if k[-1] < x <= d])
                label[d] = {1: 'expc_dec_a', 2: 'expc_dec_b', 3: 'expx_redo(empty)'}.get(pos)
    for r in rows:
        d = int(r['Dispatch_Id'])
        if not label.get(d): continue
        acc[label[d]][r['Counter_Name']].append(float(r['Counter_Value']))
        if (f, d) not in seen: seen.add((f, d)); dur[label[d]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
names = sorted({c for v in acc.values() for c in v})
print('program,dispatches,avg_us,' + ','.join(names))
for k in ('expc_sq', 'expc_dec_a', 'expc_dec_b', 'expx_redo(empty)'):
    v = acc[k]
    print(','.join([k, str(len(dur[k])), '%.1f' % (sum(dur[k]) / max(1, len(dur[k])) / 1e3)] + ['%.1f' % (sum(v[c]) / len(v[c])) if c in v else '' for c in names]))
PY
