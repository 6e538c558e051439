#!/usr/bin/env python3
"""Round 5 A/B driver: the three pairing legs of bench.py in a few seconds -- one 4096-pairing call at a time, one 65,536-pairing call, twenty 4096-pairing calls on
twenty contexts (the driver's regime) and 240 calls twelve deep -- with the shader clock and package power sampled under the saturated leg.  Results are not
checked here (tests do that).  Usage: [NBLS_LDS_LAYOUT=0] tools/pair_ab.py [tag]"""
import gzip, importlib, json, os, statistics, subprocess, sys, threading, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '22')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
tag = sys.argv[1] if len(sys.argv) > 1 else ''
pairs = json.load(gzip.open(os.path.join(ROOT, 'tests', 'golden', 'ref_vectors.json.gz')))['pairs']
g1 = b''.join(bytes.fromhex(v['g1']) for v in pairs); g2 = b''.join(bytes.fromhex(v['g2']) for v in pairs)
m = len(pairs)


def dev(n):
    G1 = (g1 * (n // m + 1))[:96 * n]; G2 = (g2 * (n // m + 1))[:192 * n]
    return torch.frombuffer(bytearray(G1), dtype=torch.uint8).cuda(), torch.frombuffer(bytearray(G2), dtype=torch.uint8).cuda()


def smi():
    try:
        o = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(o); c = next(iter(d.values()))
        sclk = [v for k, v in c.items() if 'sclk' in k.lower()]; pw = [v for k, v in c.items() if 'power' in k.lower() and 'W' in k]
        return (sclk[0] if sclk else '?'), (pw[0] if pw else '?')
    except Exception as e:   # noqa: BLE001
        return '?', repr(e)[:40]


res = {}
eng = pkg.Engine(0)
st = torch.cuda.current_stream().cuda_stream
for n, reps in ((4096, 40), (1024, 20), (2048, 20), (65536, 6)):
    d1, d2 = dev(n); out = torch.empty(576 * n, dtype=torch.uint8, device='cuda')
    for _ in range(2): eng.pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True, st)
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); eng.pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), True, st); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res['call_%d_ms' % n] = (round(min(ts), 3), round(statistics.median(ts), 3))
n = 4096; d1, d2 = dev(n)
for D, steps, reps in ((20, 20, 7), (12, 240, 3)):
    pipe = pkg.PairingPipeline(0, D)
    outs = [torch.empty(576 * n, dtype=torch.uint8, device='cuda') for _ in range(D)]
    for i in range(2 * D): pipe.submit(n, d1.data_ptr(), d2.data_ptr(), outs[pipe.slot].data_ptr(), True)
    torch.cuda.synchronize(); vs = []
    samples = []
    stop = threading.Event()
    if steps > 100:
        th = threading.Thread(target=lambda: [samples.append(smi()) for _ in range(2) if not stop.wait(0.05)]); th.start()
    for _ in range(reps * (8 if steps > 100 else 1)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps): pipe.submit(n, d1.data_ptr(), d2.data_ptr(), outs[pipe.slot].data_ptr(), True)
        torch.cuda.synchronize(); vs.append(n * steps / (time.perf_counter() - t0))
    stop.set()
    if steps > 100: th.join()
    res['inflight_%dx%d_Mps' % (D, steps)] = (round(max(vs) / 1e6, 4), round(statistics.median(vs) / 1e6, 4))
    if samples: res['smi_under_load'] = samples
    del pipe
print('PAIR_AB', tag, json.dumps(res), flush=True)
