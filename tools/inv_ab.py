#!/usr/bin/env python3
"""Round 6: the Fp inversion with one limb per lane (NBLS_INV_WIDE_MAX) against the one-lane kernel: kernel time through the timing mode of one final exponentiation at several
sizes, and the latency legs.  Usage: NBLS_INV_WIDE_MAX=<n> tools/inv_ab.py [tag]"""
import hashlib, importlib, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import oracle_py
pkg = importlib.import_module('noble-bls12-381_amd')
tag = sys.argv[1] if len(sys.argv) > 1 else ''
o = oracle_py.load(rebuild=False); eng = pkg.Engine(0)
g1, g2 = o.g1_generator(), o.g2_generator()
res = {}
for n in (1, 64, 1024, 4096, 8192, 16384, 65536):
    d1 = torch.frombuffer(bytearray(g1 * n), dtype=torch.uint8).cuda(); d2 = torch.frombuffer(bytearray(g2 * n), dtype=torch.uint8).cuda()
    out = torch.empty(576 * n, dtype=torch.uint8, device='cuda')
    call = lambda: eng.pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True)
    for _ in range(3): call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    eng.timing_enable(True); call(); torch.cuda.synchronize(); tm = eng.timing_read(); eng.timing_enable(False)
    inv = [v[0] for k, v in tm.items() if 'inv' in k]
    res[n] = {'call_ms': round(statistics.median(ts), 4), 'inv_kernel_ms': round(sum(inv), 4)}
    if n == 1:
        assert bytes(out.cpu().numpy().tobytes()) == o.pairing(g1, g2, True, False)[1]
print('INV_AB', tag, json.dumps(res), flush=True)
