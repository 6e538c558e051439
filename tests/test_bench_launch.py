"""`python bench.py --gpus N` must produce an N-rank job by itself (round-2 verdict: the flag was parsed and ignored, so the driver's
command line `python3 bench.py --gpus 8 ...` ran one rank).  Without a launcher around it bench.py re-executes itself under
torch.distributed.run with N processes; --dry-launch replaces the GPU work by a gloo rendezvous + all-reduce so that the launch path runs
on a CPU-only box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()       # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_spawns_ranks():
    d = _run(['--gpus', '2', '--dry-launch'])
    assert {k: d[k] for k in ('dry_launch', 'n_gpus', 'gpus_flag', 'rank_sum', 'backend')} == {'dry_launch': True, 'n_gpus': 2, 'gpus_flag': 2, 'rank_sum': 3, 'backend': 'gloo'}
    assert d['config3']['shards'] == [[0, 1 << 19], [1 << 19, 1 << 20]]


def test_gpus_flag_three_ranks():
    d = _run(['--gpus', '3', '--dry-launch'])
    assert d['n_gpus'] == 3 and d['rank_sum'] == 6


def test_single_rank_needs_no_launcher():
    d = _run(['--gpus', '1', '--dry-launch'])
    assert d['n_gpus'] == 1 and d['rank_sum'] == 1


def test_gpus_flag_eight_ranks():
    """the driver's scaling run: `python bench.py --gpus 8` -> eight ranks, one JSON line, every rank in the all-reduce (1 + 2 + ... + 8 = 36)"""
    d = _run(['--gpus', '8', '--dry-launch'])
    assert d['n_gpus'] == 8 and d['gpus_flag'] == 8 and d['rank_sum'] == 36
    # round 5: the legs that shard BASELINE configs[3] (2^20 independent pairings, ONE call per rank) and the one-process multi-device leg are rehearsed with the same shard
    # arithmetic, barriers and max-reduction as the real run: eight contiguous shards of 131,072 items cover the stream exactly
    c3 = d['config3']
    assert c3['pairings'] == 1 << 20 and len(c3['shards']) == 8
    assert all(hi - lo == 131072 for lo, hi in c3['shards']) and c3['shards'][0][0] == 0 and c3['shards'][-1][1] == 1 << 20
    assert all(c3['shards'][i][1] == c3['shards'][i + 1][0] for i in range(7))
    assert abs(c3['max_time_reduced'] - 0.008) < 1e-9                     # the slowest rank's time is what every rank divides by
    assert 'config3' in d['legs'] and 'multi_one_process' in d['legs']
    # round 6: the record explains itself -- one entry per rank with its own device and its own times (the real legs fill the same object and add the peer-access matrix)
    pr = d['multi_gpu']['per_rank']
    assert [e['rank'] for e in pr] == list(range(8)) and all({'local_rank', 'device', 'value_region_ms', 'config3_shard_ms', 'hw_queues'} <= set(e) for e in pr)
    assert max(e['config3_shard_ms'] for e in pr) == 8.0 and abs(c3['max_time_reduced'] * 1e3 - 8.0) < 1e-6


def test_driver_launcher_command_line():
    """the same under the launcher the driver uses for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N"""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1', '--master-port', '29611',
                        os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--dry-launch'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 4 and d['rank_sum'] == 10
