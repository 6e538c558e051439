#!/usr/bin/env python3
"""Round 6: the SWU square root of hash-to-G2 by the norm method against the Fp2 exponentiation -- hash-to-G2 alone, sign and verifyBatch with the switch
(NBLS_TUNE_H2C_NORM_MIN) at 0 and out of reach, interleaved on one box; HBM-resident inputs, medians.  Usage: tools/h2c_norm_ab.py [reps]"""
import hashlib, importlib, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module('noble-bls12-381_amd')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
eng = pkg.Engine(0)
dev = torch.device('cuda:0')


def timed(f, reps=reps):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(statistics.median(ts), 3), round(min(ts), 3)


res = {}
for n in (512, 2048, 8192):
    msgs = [hashlib.sha256(b'ab%d' % i).digest() for i in range(n)]
    row = {}
    for tag, thr in (('fp2', 1 << 30), ('norm', 0), ('fp2_again', 1 << 30), ('norm_again', 0)):
        eng.set_h2c_norm_min(thr)
        row[tag] = timed(lambda: eng.hash_to_g2_batch(msgs), 5)
    eng.set_h2c_norm_min(1 << 30); a = eng.hash_to_g2_batch(msgs); eng.set_h2c_norm_min(0); b = eng.hash_to_g2_batch(msgs)
    row['equal'] = a == b
    res['hash_to_g2_host_%d' % n] = row
    print('H2C_NORM_AB', n, json.dumps(row), flush=True)
# device-resident: sign = hash-to-G2 + ladder (nbls_sign_batch_dev), one call at a time
import random
rnd = random.Random(6)
for n in (2048, 4096, 8192, 16384, 32768, 65536):
    msgs = [hashlib.sha256(b'sg%d' % i).digest() for i in range(n)]
    blob = b''.join(msgs); offs = [0]
    for m in msgs: offs.append(offs[-1] + len(m))
    keys = b''.join((rnd.randrange(1, 1 << 254)).to_bytes(32, 'big') for _ in range(n))
    d_msg = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev); d_off = torch.tensor(offs, dtype=torch.int32).to(dev)
    d_key = torch.frombuffer(bytearray(keys), dtype=torch.uint8).to(dev)
    d_out = torch.empty(192 * n, dtype=torch.uint8, device=dev); d_st = torch.empty(n, dtype=torch.uint8, device=dev)
    row = {}
    call = lambda: eng.sign_batch_dev(n, d_msg.data_ptr(), d_off.data_ptr(), d_key.data_ptr(), d_out.data_ptr(), d_st.data_ptr())
    outs = {}
    for tag, thr in (('fp2', 1 << 30), ('norm', 0), ('fp2_again', 1 << 30), ('norm_again', 0)):
        eng.set_h2c_norm_min(thr)
        row[tag] = timed(call)
        outs[tag] = d_out.cpu().numpy().tobytes()
    row['equal'] = outs['fp2'] == outs['norm']
    res['sign_dev_%d' % n] = row
    print('H2C_NORM_AB sign_dev', n, json.dumps(row), flush=True)
eng.set_h2c_norm_min(32768)
print('H2C_NORM_AB_DONE', json.dumps(res), flush=True)
