import gzip, importlib, json, os, sys, time
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd()
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
n = int(sys.argv[1]); S = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30; own = len(sys.argv) > 4   # own: each context's own (blocking) stream instead of a torch stream
pairs = json.load(gzip.open(os.path.join(ROOT, 'tests', 'golden', 'ref_vectors.json.gz')))['pairs']
g1 = b''.join(bytes.fromhex(v['g1']) for v in pairs); g2 = b''.join(bytes.fromhex(v['g2']) for v in pairs); m = len(pairs)
G1 = (g1 * (n // m + 1))[:96 * n]; G2 = (g2 * (n // m + 1))[:192 * n]
d1 = torch.frombuffer(bytearray(G1), dtype=torch.uint8).cuda(); d2 = torch.frombuffer(bytearray(G2), dtype=torch.uint8).cuda()
engs = [pkg.Engine(0) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
outs = [torch.empty(576 * n, dtype=torch.uint8, device='cuda') for _ in range(S)]
def run(k):
    for i in range(k):
        j = i % S
        engs[j].pairing_batch_dev(n, d1.data_ptr(), d2.data_ptr(), outs[j].data_ptr(), True, None if own else streams[j].cuda_stream)
run(2 * S); torch.cuda.synchronize()
t0 = time.perf_counter(); run(reps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
same = all(bytes(outs[0][:576 * 8].cpu().numpy().tobytes()) == bytes(o[:576 * 8].cpu().numpy().tobytes()) for o in outs)
print('own-stream' if own else 'torch-stream', 'reps', reps, 'batch', n, 'streams', S, 'ms/step', round(dt / reps * 1e3, 3), 'pairings/s', round(n * reps / dt), 'consistent', same)
