import importlib, time, sys
sys.path.insert(0, '.')
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
t0 = time.perf_counter(); e = pkg.Engine(0); t1 = time.perf_counter()
es = [pkg.Engine(0) for _ in range(4)]; t2 = time.perf_counter()
print('first context %.2f s, next four %.2f s each; device memory in use %.0f MB' % (t1 - t0, (t2 - t1) / 4, (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e6))
