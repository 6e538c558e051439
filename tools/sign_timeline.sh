#!/bin/bash
# Kernel timeline of ONE sign (n keys, default 1) through nbls_sign_batch_dev: rocprofv3 --kernel-trace, the kernels of the last call with start offset, duration and queue
# (the chain hash-to-G2 -> recoding -> ladder -> inversion -> affine).  Usage: tools/sign_timeline.sh [n] [tag]
export TMPDIR=/tmp
n=${1:-1}; tag=${2:-n$n}
out=$PWD/gpurun_out/sign_timeline_$tag; rm -rf $out; mkdir -p $out
cat > $out/run.py <<PY
import hashlib, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module('noble-bls12-381_amd')
eng = pkg.Engine(0)
n = $n
sks = [(int.from_bytes(hashlib.sha256(b'tl-sk' + i.to_bytes(4, 'big')).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(n)]
msgs = [hashlib.sha256(b'tl-m' + i.to_bytes(4, 'big')).digest() for i in range(n)]
offs = np.zeros(n + 1, dtype=np.uint32); offs[1:] = np.cumsum([len(m) for m in msgs])
d_m = torch.frombuffer(bytearray(b''.join(msgs)), dtype=torch.uint8).cuda(); d_o = torch.from_numpy(offs.view(np.int32)).cuda(); d_k = torch.frombuffer(bytearray(b''.join(sks)), dtype=torch.uint8).cuda()
d_so = torch.empty(192 * n, dtype=torch.uint8, device='cuda'); d_ss = torch.empty(n, dtype=torch.uint8, device='cuda')
for i in range(4):
    torch.cuda.synchronize(); time.sleep(0.05)
    t0 = time.perf_counter(); eng.sign_batch_dev(n, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), d_so.data_ptr(), d_ss.data_ptr()); dt = time.perf_counter() - t0
print('sign_batch_dev %d: %.3f ms' % (n, dt * 1e3))
PY
rocprofv3 --kernel-trace --output-format csv -d $out -- python $out/run.py > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $out/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tail = rows[-60:]
gaps = [(int(tail[i]['Start_Timestamp']) - int(tail[i - 1]['End_Timestamp']), i) for i in range(1, len(tail))]
cut = max(gaps)[1] if gaps else 0
last = tail[cut:]
t0 = int(last[0]['Start_Timestamp'])
print('kernels of the last call: %d, span %.3f ms, sum of durations %.3f ms' % (len(last), (max(int(r['End_Timestamp']) for r in last) - t0) / 1e6, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6))
for r in last:
    print('%9.3f ms  +%7.3f ms  queue %-3s grid %-8s %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('Queue_Id', '?'), r.get('Grid_Size', r.get('Grid_Size_X', '?')), r['Kernel_Name'][:60]))
PY
grep sign_batch_dev $out/run.log | tee -a $out/timeline.txt
