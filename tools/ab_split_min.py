#!/usr/bin/env python3
"""Round 6 A/B: where the fused Miller program stops winning for ONE call at a time.  profiles/round6_ab_ls2_4096.txt showed a 4096-pairing call at 2.18 ms with the fused program
against 2.31 ms with LINES + ACC (the default from 4096 pairs since round 4).  Times pairing_batch_dev at several sizes with the split threshold below / above the size, interleaved."""
import gzip, importlib, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
pairs = json.load(gzip.open(os.path.join(ROOT, 'tests', 'golden', 'ref_vectors.json.gz')))['pairs']
g1 = b''.join(bytes.fromhex(v['g1']) for v in pairs); g2 = b''.join(bytes.fromhex(v['g2']) for v in pairs); m = len(pairs)
eng = pkg.Engine(0); st = torch.cuda.current_stream().cuda_stream
sizes = [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '3072,4096,4608,5120,6144,7168,8191'.split(','))]
for n in sizes:
    d1 = torch.frombuffer(bytearray((g1 * (n // m + 1))[:96 * n]), dtype=torch.uint8).cuda(); d2 = torch.frombuffer(bytearray((g2 * (n // m + 1))[:192 * n]), dtype=torch.uint8).cuda()
    out = torch.empty(576 * n, dtype=torch.uint8, device='cuda'); res = {}
    outs = {}
    for rep in range(2):
        for mode, thr in (('split', 1), ('fused', 1 << 30)):
            eng.set_split_miller_min(thr)
            for _ in range(3): eng.pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True, st)
            torch.cuda.synchronize(); ts = []
            for _ in range(30):
                t0 = time.perf_counter(); eng.pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True, st); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            res.setdefault(mode, []).append(round(statistics.median(ts), 3))
            outs[mode] = bytes(out.cpu().numpy().tobytes())
    print('SPLIT_AB n=%d' % n, res, 'same_bytes', outs['split'] == outs['fused'], flush=True)
