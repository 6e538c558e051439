#!/usr/bin/env python3
"""Round 5 experiment: the driver's regime -- a burst of 20 calls of 4096 pairings on 20 pool contexts, timed from the first submit to the device being idle -- under variations of
what the contexts run (chained / unchained final exponentiation, fused / split Miller loop on a share of the contexts, depth).  Best and median M pairings/s of `reps` bursts each."""
import gzip, importlib, json, os, statistics, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '22')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
pairs = json.load(gzip.open(os.path.join(ROOT, 'tests', 'golden', 'ref_vectors.json.gz')))['pairs']
g1 = b''.join(bytes.fromhex(v['g1']) for v in pairs); g2 = b''.join(bytes.fromhex(v['g2']) for v in pairs)
n = 4096; m = len(pairs)
d1 = torch.frombuffer(bytearray((g1 * (n // m + 1))[:96 * n]), dtype=torch.uint8).cuda(); d2 = torch.frombuffer(bytearray((g2 * (n // m + 1))[:192 * n]), dtype=torch.uint8).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9


def burst(pipe, steps, outs, warm=5):
    for i in range(warm): pipe.submit(n, d1.data_ptr(), d2.data_ptr(), outs[pipe.slot].data_ptr(), True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): pipe.submit(n, d1.data_ptr(), d2.data_ptr(), outs[pipe.slot].data_ptr(), True)
    torch.cuda.synchronize()
    return n * steps / (time.perf_counter() - t0) / 1e6


def run(tag, D, steps, setup):
    pipe = pkg.PairingPipeline(0, D)
    setup(pipe)
    outs = [torch.empty(576 * n, dtype=torch.uint8, device='cuda') for _ in range(D)]
    for i in range(2 * D): pipe.submit(n, d1.data_ptr(), d2.data_ptr(), outs[pipe.slot].data_ptr(), True)
    torch.cuda.synchronize()
    v = [burst(pipe, steps, outs) for _ in range(reps)]
    print('BURST %-44s D=%2d steps=%d: best %.4f median %.4f M pairings/s' % (tag, D, steps, max(v), statistics.median(v)), flush=True)
    pipe.close()


def some(pipe, frac, f):
    k = int(len(pipe.engines) * frac)
    for e in pipe.engines[:k]: f(e)


if len(sys.argv) > 2 and sys.argv[2] == 'depth':
    for rep in range(2):
        for D in (4, 5, 7, 8, 10, 12, 14, 16, 20):
            run('pool default', D, 20, lambda p: None)
    for D in (8, 10, 12, 16, 20):
        run('pool default, 240 steps', D, 240, lambda p: None)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == 'd12':
    for rep in range(3):
        run('pool default', 12, 20, lambda p: None)
        run('all chained', 12, 20, lambda p: some(p, 1.0, lambda e: e.set_chain_max(8192)))
        run('fused Miller below 4096', 12, 20, lambda p: some(p, 1.0, lambda e: e.set_split_miller_min(4097)))
        run('pool default', 13, 20, lambda p: None)
        run('pool default', 11, 20, lambda p: None)
    sys.exit(0)
for rep in range(2):
    run('pool default (split Miller, unchained)', 20, 20, lambda p: None)
    run('all chained', 20, 20, lambda p: some(p, 1.0, lambda e: e.set_chain_max(8192)))
    run('half chained', 20, 20, lambda p: some(p, 0.5, lambda e: e.set_chain_max(8192)))
    run('quarter chained', 20, 20, lambda p: some(p, 0.25, lambda e: e.set_chain_max(8192)))
    run('half fused Miller', 20, 20, lambda p: some(p, 0.5, lambda e: e.set_split_miller_min(1 << 30)))
    run('quarter fused Miller', 20, 20, lambda p: some(p, 0.25, lambda e: e.set_split_miller_min(1 << 30)))
    run('depth 10 (two calls per stream)', 10, 20, lambda p: None)
    run('depth 22', 22, 20, lambda p: None)
