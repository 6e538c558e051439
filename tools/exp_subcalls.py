#!/usr/bin/env python3
"""Experiment: ONE call split into K concurrent sub-calls on K contexts of the same GPU (would the library gain from doing that internally?).
Measured: 65,536 pairings 28.0 -> 26.4 ms with two sub-batches (two-program Miller loop in each); verifyBatch of 65,536 signatures 27.6 -> 34.0 / 31.4 / 29.8 ms with
2 / 3 / 4 shards (per-shard overheads and the lock-step phases outweigh the overlap) -- independent calls in flight are what pays (bench.py).  Usage: tools/exp_subcalls.py"""
import hashlib, importlib, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
os.environ.setdefault('GPU_MAX_HW_QUEUES','8')
import torch
pkg = importlib.import_module('noble-bls12-381_amd'); par = importlib.import_module('noble-bls12-381_amd.parallel')
import gzip, json
pairs = json.load(gzip.open(ROOT+'/tests/golden/ref_vectors.json.gz'))['pairs']
n=65536
g1=b''.join(bytes.fromhex(v['g1']) for v in pairs); g2=b''.join(bytes.fromhex(v['g2']) for v in pairs); m=len(pairs)
G1=(g1*(n//m+1))[:96*n]; G2=(g2*(n//m+1))[:192*n]
d1=torch.frombuffer(bytearray(G1),dtype=torch.uint8).cuda(); d2=torch.frombuffer(bytearray(G2),dtype=torch.uint8).cuda(); out=torch.empty(576*n,dtype=torch.uint8,device='cuda')
engs=[pkg.Engine(0) for _ in range(4)]
def run_split(K, split_min):
    for e in engs: e.set_split_miller_min(split_min)
    bounds=[(n*k//K, n*(k+1)//K) for k in range(K)]
    def once():
        for k,(lo,hi) in enumerate(bounds):
            engs[k].pairing_batch_dev(hi-lo, d1.data_ptr()+96*lo, d2.data_ptr()+192*lo, out.data_ptr()+576*lo, True, None)
    once(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): once(); torch.cuda.synchronize()
    return (time.perf_counter()-t0)/5*1e3
print('pairing 65536 one call: %.2f ms' % run_split(1, 49152))
for K in (2,3,4):
    print('pairing 65536 as %d sub-batches on %d streams: split_min=0 %.2f ms, default %.2f ms' % (K, K, run_split(K, 0), run_split(K, 49152)))
# verifyBatch sharded over K contexts of one GPU with host threads
import oracle_py
oracle = oracle_py.load()
eng=engs[0]
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
sks = [(int.from_bytes(hashlib.sha256(b'k' + i.to_bytes(4, 'big')).digest(), 'big') % (R - 1) + 1).to_bytes(32, 'big') for i in range(n)]
msgs = [hashlib.sha256(b'm' + i.to_bytes(4, 'big')).digest() for i in range(n)]
pks = eng.get_public_keys(sks); aff, st = eng.sign_batch_affine(msgs, sks); agg, z = eng.point_sum(aff, g2=True); sig = eng.compress_g2(agg)
uni = b''.join(oracle.expand_message_xmd(mm, oracle_py.DST_DEFAULT, 256) for mm in msgs)
d_sig = torch.frombuffer(bytearray(sig), dtype=torch.uint8).cuda(); d_uni = torch.frombuffer(bytearray(uni), dtype=torch.uint8).cuda(); d_pk = torch.frombuffer(bytearray(b''.join(pks)), dtype=torch.uint8).cuda()
for e in engs: e.set_split_miller_min(49152)
assert eng.verify_batch_dev(n, d_sig.data_ptr(), d_uni.data_ptr(), d_pk.data_ptr()) is True
t0=time.perf_counter()
for _ in range(3): eng.verify_batch_dev(n, d_sig.data_ptr(), d_uni.data_ptr(), d_pk.data_ptr())
print('verifyBatch one call: %.2f ms' % ((time.perf_counter()-t0)/3*1e3))
for K in (2,3,4):
    bounds=[(n*k//K, n*(k+1)//K) for k in range(K)]
    parts=[torch.zeros(576,dtype=torch.uint8,device='cuda') for _ in range(K)]
    streams=[torch.cuda.Stream() for _ in range(K)]
    def shard(k):
        lo,hi=bounds[k]
        with torch.cuda.stream(streams[k]):
            engs[k].verify_batch_partial_dev(hi-lo, d_sig.data_ptr() if k==0 else None, d_uni.data_ptr()+256*lo, d_pk.data_ptr()+48*lo, parts[k].data_ptr(), stream=streams[k].cuda_stream)
    def once():
        ths=[threading.Thread(target=shard,args=(k,)) for k in range(K)]
        for t in ths: t.start()
        for t in ths: t.join()
        for s_ in streams: s_.synchronize()
        allp=torch.cat(parts); o=torch.empty(576,dtype=torch.uint8,device='cuda')
        engs[0].fp12_product_final_dev(K, allp.data_ptr(), o.data_ptr(), final_exp=True, stream=None); torch.cuda.synchronize()
        return bytes(o.cpu().numpy().tobytes())
    one=bytes(47)+b'\x01'+bytes(528)
    assert once()==one
    t0=time.perf_counter()
    for _ in range(3): once()
    print('verifyBatch as %d shards (threads, contexts of one GPU): %.2f ms' % (K,(time.perf_counter()-t0)/3*1e3))
