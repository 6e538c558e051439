import importlib, os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import goldenio
from goldenio import hx
pkg = importlib.import_module('noble-bls12-381_amd')
eng = pkg.Engine(0)
g = goldenio.load('ref_vectors.json.gz'); vb = g['verify_batch']
msgs, pks = [hx(m) for m in vb['msgs']], [hx(p) for p in vb['pks']]
for i in range(3): print(eng.verify_batch(hx(vb['agg_sig']), msgs, pks))
