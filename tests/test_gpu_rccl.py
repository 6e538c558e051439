"""The RCCL code path of the one-process-per-GPU form (parallel.py) executed for real: torch.distributed with backend "nccl" (= RCCL on ROCm), the
all-gather of 576-byte Fp12 partials on uint8 device tensors and the int32 flag all-reduce -- with ONE rank, which is all a one-GPU box allows
(NBLS_FORCE_COLLECTIVES=1 makes parallel.py run the collectives although the world size is 1).  What it pins: process-group creation with
device_id, the import order torch -> libnbls.so, HSA_ENABLE_IPC_MODE_LEGACY=0, dtype / device of the exchanged tensors, and that the
results equal the single-context ones bit for bit.  World size 2 is covered on CPU with gloo (test_distributed_cpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import importlib, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', device_id=torch.device('cuda', 0))
pkg = importlib.import_module('noble-bls12-381_amd')
par = importlib.import_module('noble-bls12-381_amd.parallel')
import goldenio, oracle_py
from goldenio import hx
golden = goldenio.load('ref_vectors.json.gz'); oracle = oracle_py.load(rebuild=False)
eng = pkg.Engine(0); be = par.EngineBackend(eng)
dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
# product: all-gather of one partial, local product, shared final exponentiation
pairs = golden['pairs'][:24]
g1 = b''.join(hx(v['g1']) for v in pairs); g2 = b''.join(hx(v['g2']) for v in pairs)
res = par.miller_product_sharded(be, dev(g1), dev(g2), final_exp=True)
assert bytes(res.cpu().numpy().tobytes()) == oracle.miller_product(g1, g2, True), 'product'
# verifyBatch: flag all-reduce + all-gather
vb = golden['verify_batch']
sig, msgs, pks = hx(vb['agg_sig']), [hx(m) for m in vb['msgs']], [hx(p) for p in vb['pks']]
uni = [oracle.expand_message_xmd(m, oracle_py.DST_DEFAULT, 256) for m in msgs]
assert par.verify_batch_sharded(be, dev(sig), dev(b''.join(uni)), dev(b''.join(pks))) is True
bad = list(uni); bad[0] = oracle.expand_message_xmd(b'another message', oracle_py.DST_DEFAULT, 256)
assert par.verify_batch_sharded(be, dev(sig), dev(b''.join(bad)), dev(b''.join(pks))) is False
# barrier + max all-reduce as bench.py times its steps
dist.barrier(); t = torch.tensor([1.5], device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t.item()) == 1.5
dist.destroy_process_group()
print('rccl ok')
'''


def test_rccl_collectives_one_rank():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', NBLS_FORCE_COLLECTIVES='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-c', SCRIPT % {'root': ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'rccl ok' in out.stdout, out.stdout + out.stderr


def test_bench_distributed_branches_one_rank():
    """bench.py's N > 1 branches (process group, barriers, max all-reduce of the timings, sharded product and verifyBatch legs with their all-gathers)
    run with one rank through --force-dist: the driver's multi-GPU launch executes exactly these lines."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_PORT='29537')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--steps', '4', '--warmup', '1', '--batch', '256', '--inflight', '2', '--verify-batch', '64',
           '--product-terms', '256', '--sign-batch', '0', '--msm-points', '0', '--large-batch', '0', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])      # RCCL prints its version banner on stdout too
    assert line['rccl_ranks'] == 1 and line['n_gpus'] == 1 and line['product']['result_is_one'] and line['verify_batch']['n_signatures'] == 64


def test_bench_two_ranks_self_launched():
    """`python bench.py --gpus 2` with no launcher around it, on the GPU box: bench.py starts two ranks under torch.distributed.run, each with its own HIP
    contexts and three streams; the ranks synthesise their own inputs (seeded by rank), run the timed steps between barriers, shard the
    product and verifyBatch legs (all-gather of two 576-byte partials + flag all-reduce), and rank 0 prints ONE line with n_gpus = 2.  RCCL refuses two
    ranks on one device, so on a one-GPU box the ranks share GPU 0 and exchange through gloo (--dist-backend gloo); with two or more GPUs the same test
    runs over RCCL, one rank per device."""
    import json
    import torch
    backend = 'nccl' if torch.cuda.device_count() >= 2 else 'gloo'
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', backend, '--steps', '6', '--warmup', '2', '--batch', '512', '--inflight', '3',
           '--verify-batch', '128', '--product-terms', '512', '--sign-batch', '0', '--msm-points', '0', '--large-batch', '0', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-3000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['dist_backend'] == backend and line['value'] > 0
    assert line['product']['result_is_one'] and '2 x 576' in line['product']['exchange']
    assert line['verify_batch']['n_signatures'] == 128 and 'sharded over 2 rank' in line['verify_batch']['note']
