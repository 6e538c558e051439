import importlib, os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
import oracle_py
oracle = oracle_py.load(rebuild=False)
sys.argv = ['x']
import bench
G1, G2 = bench.synth_points(oracle, 4096)
d1 = torch.frombuffer(bytearray(G1), dtype=torch.uint8).cuda(); d2 = torch.frombuffer(bytearray(G2), dtype=torch.uint8).cuda()
for D in (10, 12):
    pipe = pkg.PairingPipeline(0, D)
    outs = [torch.empty(576 * 4096, dtype=torch.uint8, device='cuda') for _ in range(D)]
    for i in range(2 * D): pipe.submit(4096, d1.data_ptr(), d2.data_ptr(), outs[pipe.slot].data_ptr(), True)
    torch.cuda.synchronize()
    for K in (20, 20, 60):
        t0 = time.perf_counter()
        for i in range(K): pipe.submit(4096, d1.data_ptr(), d2.data_ptr(), outs[pipe.slot].data_ptr(), True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('D', D, 'K', K, 'submit %.3f ms' % ((t1 - t0) * 1e3), 'total %.3f ms' % ((t2 - t0) * 1e3), 'M/s %.3f' % (K * 4096 / (t2 - t0) / 1e6))
