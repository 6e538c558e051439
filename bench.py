#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X pairing engine.

One "step" = one pass of the hot path over one batch: pairing(P_i, Q_i) for a batch of 4096 independent, pre-validated
point pairs per GPU (BASELINE.json configs[1]; Miller loop with on-the-fly line computation + final exponentiation),
inputs and outputs resident in HBM, called through the C ABI (libnbls.so).  Consecutive steps are independent batches; they
are submitted to `--inflight` engine contexts (default 12), each with its own HIP stream and scratch, so that they overlap
on the GPU the way a service keeps several requests in flight (a 4096-pairing call alone fills the chip one wavefront
deep).  `value` is the throughput of the K timed steps; `single_stream` reports the strictly serial figure (= per-batch
latency) next to it, and the roofline object is measured on one batch running alone.  Multi-GPU: one process per GPU,
batches sharded with no data-path collective (weak scaling), barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line (see DESIGN.md section 6 for the field definitions):
  value        pairings/s, whole job
  roofline     dominant kernel (nbls_vm_kernel) against the gfx950 integer-multiplier issue rate:
               achieved = algorithmic 32x32 multiply-adds per second (Fp multiplications of the
               reference algorithm x 300 MAD32) from HIP-event kernel durations measured in this run;
               peak = 256 CU x 64 lanes/clk (one 32x32->64 multiply-add per lane per 4 clocks per SIMD, measured with
               tools/ubench/int_rates.hip) x 2.4 GHz
  cpu_baseline oracle/ (C restatement of the reference algorithm) on the host cores, bounded sample, rank 0, N=1 only
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

BATCH = 4096
FPMUL_MILLER = 7556        # Fp mul+sqr of calcPairingPrecomputes + millerLoop
FPMUL_FINALEXP = 12166     # Fp mul+sqr of finalExponentiate
MAD_PER_FPMUL = 300        # 12-limb CIOS Montgomery product: 2*12^2 + 12
PEAK_TMAD = 256 * 64 * 2.4e9 / 1e12   # T MAD32/s: 256 CU x 64 lanes/clk/CU x 2.4 GHz (measured issue rate: tools/ubench)
HBM_PEAK_GBPS = 8000.0


R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def bench_scalar(seed, k):
    """s_k = SHA-256("nbls-bench-v1" || seed || u64be(k)) mod r, zero rejected (the next counter value in its place would be the survey's rule; SHA-256 never gave one)"""
    s = int.from_bytes(hashlib.sha256(b'nbls-bench-v1' + seed.to_bytes(4, 'big') + k.to_bytes(8, 'big')).digest(), 'big') % R_ORDER
    assert s != 0
    return s


def synth_points(oracle, n, seed=0x6e626c73):
    """Deterministic valid (G1, G2) pairs for the SECONDARY legs (product, aggregation): 64 distinct random multiples of the generators made by the oracle's
    scalar multiplication (host, setup only) and combined into n distinct pairs (P_a, Q_b).  The pairing legs use synth_stream."""
    g1, g2 = oracle.g1_generator(), oracle.g2_generator()
    P, Q = [], []
    for i in range(64):
        a = int.from_bytes(hashlib.sha256(b'nbls-bench-v1' + seed.to_bytes(4, 'big') + (2 * i).to_bytes(8, 'big')).digest(), 'big') % (2 ** 254) + 1
        b = int.from_bytes(hashlib.sha256(b'nbls-bench-v1' + seed.to_bytes(4, 'big') + (2 * i + 1).to_bytes(8, 'big')).digest(), 'big') % (2 ** 254) + 1
        P.append(oracle.g1_mul(g1, a)[1])
        Q.append(oracle.g2_mul(g2, b)[1])
    G1 = b''.join(P[i % 64] for i in range(n))
    G2 = b''.join(Q[(i // 64 + 3 * i) % 64] for i in range(n))
    return G1, G2


def synth_stream(eng, oracle, n, seed=0x6e626c73, first=0, check=(0, 1, -1)):
    """SURVEY.md 8(d)'s per-item stream: pair i is (P_i, Q_i) = ([s_2i] G1, [s_2i+1] G2), every pair distinct and uniformly distributed over the subgroups.  The 2 n scalar
    multiplications run on the GPU (the constant-time ladders of getPublicKey / sign: setup, never timed); the pairs at `check` are compared with the oracle's own
    multiplications of the same scalars.  first: index of the first item (a rank's shard of a larger stream)."""
    ks1 = [bench_scalar(seed, 2 * (first + i)).to_bytes(32, 'big') for i in range(n)]
    ks2 = [bench_scalar(seed, 2 * (first + i) + 1).to_bytes(32, 'big') for i in range(n)]
    G1, st1 = eng.point_mul_batch(ks1)
    G2, st2 = eng.point_mul_batch(ks2, pts=oracle.g2_generator() * n, g2=True)
    assert not any(st1) and not any(st2), 'a bench scalar gave the zero point'
    for c in check:
        i = c % n
        assert G1[96 * i:96 * i + 96] == oracle.g1_mul(oracle.g1_generator(), int.from_bytes(ks1[i], 'big'))[1], 'bench input check (G1) against the oracle failed'
        assert G2[192 * i:192 * i + 192] == oracle.g2_mul(oracle.g2_generator(), int.from_bytes(ks2[i], 'big'))[1], 'bench input check (G2) against the oracle failed'
    return G1, G2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=512, help='timed steps (default: about one second of GPU time)')
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--inflight', type=int, default=None, help='batches kept in flight on separate HIP streams during the timed steps (1 = strictly one batch at a time)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='run the torch.distributed / RCCL code paths (process group, barriers, all-gathers, sharded legs) even with one rank: the only way to execute them on a one-GPU box')
    ap.add_argument('--verify-batch', type=int, default=65536, help='signatures in the verifyBatch leg (BASELINE configs[2]); 0 disables')
    ap.add_argument('--verify-inflight', type=int, default=3, help='verifyBatch calls kept in flight (one context and host thread each) in the pipelined part of the verifyBatch leg')
    ap.add_argument('--verify-sharded', action='store_true', help='run the multi-GPU form of the verifyBatch leg (parallel.verify_batch_sharded) even on one rank')
    ap.add_argument('--msm-points', type=int, default=65536, help='points in the multi-scalar multiplication leg (SURVEY 8(f).3); 0 disables')
    ap.add_argument('--sign-batch', type=int, default=8192, help='signatures produced in the sign leg (SURVEY 8(f).1); 0 disables')
    ap.add_argument('--large-batch', type=int, default=65536, help='pairings of the saturated single-call leg (roofline at a launch that fills every SIMD three wavefronts deep); 0 disables')
    ap.add_argument('--product-terms', type=int, default=262144, help='terms of the sharded multi-pairing product leg (BASELINE configs[4]); 0 disables')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'], help='torch.distributed backend of an N > 1 run: nccl (= RCCL, the real thing) or gloo, which lets several ranks share one GPU (rank r uses device r mod device count) so that the whole N-rank job -- launcher, sharded legs, barriers -- runs on a one-GPU box (RCCL refuses two ranks on one device); tests/test_gpu_rccl.py')
    ap.add_argument('--mark-timed-region', action='store_true', help='bracket the timed steps with two tiny torch fill kernels, so that a rocprofv3 kernel trace of the run shows where the timed region starts and ends (tools/profile_round3.sh)')
    ap.add_argument('--config3-pairings', type=int, default=1 << 20, help='independent pairings of the BASELINE configs[3] leg, sharded over the ranks, ONE call per rank (1M over 8 GPUs = 131,072 per rank); 0 disables')
    ap.add_argument('--dry-launch', action='store_true', help='launch check without a GPU: the ranks of --gpus N rendezvous over gloo, all-reduce their ranks and rank 0 prints one JSON line (tests/test_bench_launch.py)')
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU under torch.distributed.run, the
    # same command line the driver uses) and let rank 0 of that job print the JSON line.  Under torchrun (WORLD_SIZE set) this is skipped.
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
        env = dict(os.environ); env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.dry_launch:
        import torch
        import torch.distributed as dist
        world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29519'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group(backend='gloo')
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        # rehearsal of the legs that shard BASELINE configs[3] / [2] over the ranks: the same shard arithmetic and the same order of barriers / reductions as the real run
        par = importlib.import_module('noble-bls12-381_amd.parallel')
        lo3, hi3 = par.shard_bounds(args.config3_pairings, world, rank)
        shards = [None] * world
        dist.all_gather_object(shards, (lo3, hi3))
        dist.barrier()
        t3 = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t3, op=dist.ReduceOp.MAX)
        dist.barrier()
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {'rank': rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'device': 'cpu (dry launch)', 'value_region_ms': 1.0 + rank, 'config3_shard_ms': 1000.0 * 0.001 * (rank + 1), 'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES')})
        if rank == 0:
            print(json.dumps({'multi_gpu': {'per_rank': per_rank, 'rccl_ranks': None, 'backend': 'gloo', 'visible_devices': 0, 'peer_access': None}, 'dry_launch': True, 'n_gpus': world, 'gpus_flag': args.gpus, 'rank_sum': int(t.item()), 'backend': 'gloo',
                              'config3': {'pairings': args.config3_pairings, 'shards': shards, 'max_time_reduced': float(t3.item())},
                              'legs': ['value', 'product', 'verify_batch_sharded', 'config3'] + (['multi_one_process'] if world > 1 else [])}), flush=True)
        dist.destroy_process_group()
        return

    # the HIP runtime maps streams onto this many hardware queues (default 4): with fewer queues than batches in flight two
    # streams share a queue and their kernels serialise (must be set before the runtime initialises)
    # One hardware queue per stream in flight (twelve pool contexts + the streams of the other legs) and no more than 22 in all: from ~24 user queues the process oversubscribes the device's hardware queue
    # slots, and the legs that create further streams after the in-flight contexts (verifyBatch, sign) then pay a queue switch per launch -- with 32: verifyBatch 29.6 instead of
    # 23.4 ms, one verify 7.7 instead of 3.7 ms, sign 17.4 instead of 6.3 ms; with 20 the twenty streams share with the default stream and lose 4 % (tools/ab_queues20.sh,
    # profiles/round4_ab_queues20.txt).
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '22')
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the engine has no CPU path)')
    if args.dist_backend == 'gloo':
        local_rank = local_rank % torch.cuda.device_count()      # ranks may share a GPU under gloo
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist      # every collective below is gated on this
    if multi:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.force_dist:
            for k, v in (('MASTER_PORT', '29517'), ('RANK', '0'), ('WORLD_SIZE', '1'), ('NBLS_FORCE_COLLECTIVES', '1')): os.environ.setdefault(k, v)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.dist_backend == 'gloo':
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    pkg = importlib.import_module('noble-bls12-381_amd')
    import oracle_py
    ranks_seen = None
    if multi:   # a real collective over the job's backend before anything is timed: how many distinct ranks answered (the SCALE record can check that RCCL saw N of them)
        dev_ = 'cpu' if args.dist_backend == 'gloo' else 'cuda'     # gloo gathers host tensors (ranks may share a GPU there)
        mine = torch.tensor([rank], dtype=torch.int32, device=dev_); everyone = torch.empty(world, dtype=torch.int32, device=dev_)
        dist.all_gather_into_tensor(everyone, mine)
        ranks_seen = len(set(everyone.cpu().tolist()))
    oracle = oracle_py.load(rebuild=not os.path.exists(os.path.join(ROOT, 'oracle', 'libnbls_oracle.so')))
    # calls in flight: twelve contexts, whatever the number of steps (an explicit --inflight is taken as given).  Round 4 gave the driver's 20 steps twenty streams (one batch each:
    # 2.86 against 2.79 M pairings/s for 10 x 2 then); with this round's build the order is the other way round -- 20 steps on 10 / 12 / 14 / 20 contexts: 2.95 / 2.97 / 2.97 / 2.93 M
    # (medians of two interleaved sweeps, tools/burst_ab.py, profiles/round5_burst_depth.txt): streams that carry two batches are no longer phase-locked with their neighbours
    D = max(1, args.inflight if args.inflight is not None else 12)
    pipe = pkg.PairingPipeline(local_rank, D)     # D engine contexts, each with its own stream and scratch (noble-bls12-381_amd/pipeline.py)
    eng = pipe.engines[0]

    n = args.batch
    G1, G2 = synth_stream(eng, oracle, n, seed=0x6e626c73, first=rank * n)      # item i of rank r is pair r * n + i of the stream
    d_g1 = torch.frombuffer(bytearray(G1), dtype=torch.uint8).cuda()
    d_g2 = torch.frombuffer(bytearray(G2), dtype=torch.uint8).cuda()
    d_outs = [torch.empty(576 * n, dtype=torch.uint8, device='cuda') for _ in range(D)]
    d_out = d_outs[0]
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()

    def step(i):     # one pass of the hot path over one batch; consecutive steps go to different streams and overlap on the GPU
        pipe.submit(n, d_g1.data_ptr(), d_g2.data_ptr(), d_outs[pipe.slot].data_ptr(), True)

    def step1():     # the same step, strictly one batch at a time (roofline leg and the single-stream figure)
        eng.pairing_batch_dev(n, d_g1.data_ptr(), d_g2.data_ptr(), d_out.data_ptr(), True, stream)

    # parity spot check inside the bench: first 8 results of every in-flight buffer against the oracle.  The oracle's reference is computed BEFORE the GPU runs (round 5): the
    # device then goes from these calls through the warm-up steps into the timed steps without tens of milliseconds of idling in between (a burst that starts on a chip that has just
    # dropped to its idle clocks measures the clock ramp, not the engine: tools/ab_bench_depth.sh)
    ref, _ = oracle.pairing_batch(G1[:96 * 8], G2[:192 * 8], True, False, threads=8)
    ref_first = os.environ.get('NBLS_BENCH_REF_FIRST', '1') != '0'
    for i in range(D):
        step(i)
    torch.cuda.synchronize()
    if not ref_first:
        ref, _ = oracle.pairing_batch(G1[:96 * 8], G2[:192 * 8], True, False, threads=8)
    heads = torch.stack([o[:576 * 8] for o in d_outs]).cpu().numpy()      # one copy for all in-flight buffers
    for k in range(D):
        assert bytes(heads[k].tobytes()) == ref, 'bench parity check failed'

    local_times = []      # this rank's own time of every timed region (the line's `value` uses the maximum over the ranks; the per-rank figures go into `multi_gpu`)

    def timed_region(mark_it):
        """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize on both sides; the maximum over the ranks"""
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        if mark_it:
            mark = torch.empty(3, dtype=torch.float32, device='cuda'); mark.fill_(1.0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        d_ = time.perf_counter() - t0
        local_times.append(d_)
        if mark_it:
            mark.fill_(2.0); torch.cuda.synchronize()
        if multi:
            t = torch.tensor([d_], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d_ = float(t.item())
        return d_

    # `value` = the driver's protocol as it comes: W untimed warm-up steps, then exactly K timed steps (rounds 1-4; round 5 had put 8 D untimed "pre-warm" steps in front of it, which the
    # round-5 review and ADVICE.md rightly called a change of method: +6.1 % for the same burst, profiles/round5_ab_bench_prewarm.txt).  The chip drops to its idle power state while the
    # host checks the parity bytes and W = 5 steps (7 ms of work) do not lift it back, so at the driver's 20 steps `value` includes the clock ramp.  The figure on a chip that has been
    # busy for a tenth of a second -- what a service sees -- is reported separately as `steady_state` (PREWARM untimed steps, then the same W + K); with hundreds of timed steps the
    # ramp is noise and the second region is skipped.  NBLS_BENCH_PREWARM=0 switches the second region off.
    PREWARM = int(os.environ.get('NBLS_BENCH_PREWARM', str(8 * D)))
    dt = timed_region(args.mark_timed_region)
    dt_steady = None
    if PREWARM > 0 and args.steps <= 64:
        for i in range(PREWARM):
            step(i)
        dt_steady = timed_region(False)
    # strictly serial figure (one batch at a time on one stream) = per-batch latency, this rank
    torch.cuda.synchronize()
    eng.set_chain_max(8192)             # ... and the chained final exponentiation (6 launches per call; the in-flight contexts run it as seven launches, pipeline.py)
    eng.set_inv_wide_max(4096)          # ... and the library's default for the inversion (nbls_pool_init set 256 on the in-flight contexts)
    eng.set_ls_max(1024, 2048)          # ... and for the lane-split forms (0 / 0 on the in-flight contexts)
    eng.set_split_miller_min(4097)      # the single-call legs use the library's default choice of Miller programs (SPLIT_MILLER_MIN in csrc/nbls_internal.h: the fused program up to 4096 pairs; the in-flight contexts were set to 0)
    serial_steps = max(8, min(args.steps, 320))
    s0 = time.perf_counter()
    for _ in range(serial_steps):
        step1()
    torch.cuda.synchronize()
    dt_serial = (time.perf_counter() - s0) * args.steps / serial_steps     # scaled to args.steps (the figures below divide by it)
    step = step1
    total = n * args.steps * world
    value = total / dt

    # ---- secondary leg (all N): 2^18-term multi-pairing product with shared final exponentiation, terms sharded over the
    # ranks, ONE all-gather of 576-byte Fp12 partials over RCCL (BASELINE configs[4])
    product = None
    try:   # a failing secondary leg must not cost the headline line (its error is reported in its place)
        if args.product_terms > 0:
            par = importlib.import_module('noble-bls12-381_amd.parallel')
            lo, hi = par.shard_bounds(args.product_terms, world, rank)
            m = hi - lo
            # interleave (P, Q) and (-P, Q): the product is ONE by bilinearity, a size-independent parity check
            P1, Q1 = synth_points(oracle, 64, seed=77)
            negP = b''.join(oracle.un('g1_neg_aff', P1[96 * i:96 * i + 96], 96) for i in range(64))
            pg1 = bytearray(); pg2 = bytearray()
            for j in range(0, 128, 2):
                i = j // 2
                pg1 += P1[96 * i:96 * i + 96] + negP[96 * i:96 * i + 96]
                pg2 += Q1[192 * i:192 * i + 192] * 2
            reps_needed = (m + 127) // 128
            t1 = torch.frombuffer(bytearray(bytes(pg1) * reps_needed)[:96 * m], dtype=torch.uint8).cuda()
            t2 = torch.frombuffer(bytearray(bytes(pg2) * reps_needed)[:192 * m], dtype=torch.uint8).cuda()
            be = par.EngineBackend(eng)
            res = par.miller_product_sharded(be, t1, t2)
            torch.cuda.synchronize()
            one = bytes(47) + b'\x01' + bytes(528)
            assert (m % 2 == 0) and bytes(res.cpu().numpy().tobytes()) == one, 'product parity check failed'
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            p0 = time.perf_counter()
            preps = 3
            for _ in range(preps):
                res = par.miller_product_sharded(be, t1, t2)
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            pdt = time.perf_counter() - p0
            if multi:
                t = torch.tensor([pdt], dtype=torch.float64, device='cuda')
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                pdt = float(t.item())
            product = {'metric': 'multi-pairing product terms/sec (shared final exponentiation)', 'value': round(args.product_terms * preps / pdt, 2), 'terms': args.product_terms,
                       'ms_per_product': round(pdt / preps * 1e3, 3), 'exchange': 'all-gather of %d x 576 B Fp12 partials' % world if world > 1 else 'none (1 rank)', 'result_is_one': True}
            del t1, t2
    except Exception as e:   # noqa: BLE001
        if multi:
            raise     # the leg contains collectives: a rank that skipped them would leave its peers waiting; with several ranks the job fails as a whole
        product = {'error': repr(e)}

    # ---- secondary leg (N > 1): verifyBatch of --verify-batch signatures with the (key, message) pairs sharded over the ranks
    # (BASELINE configs[2] at 2 / 4 / 8 GPUs): every rank decodes + hashes its shard and reduces it to one Fp12 partial, ONE all-gather
    # of 576-byte partials, shared final exponentiation (parallel.verify_batch_sharded).  Setup (keys, signatures) also runs on the GPUs.
    vshard = None
    try:   # a failing secondary leg must not cost the headline line (its error is reported in its place)
        if args.verify_batch > 0 and (multi or args.verify_sharded):
            par = importlib.import_module('noble-bls12-381_amd.parallel')
            nv = args.verify_batch
            lo, hi = par.shard_bounds(nv, world, rank)
            sks_l = [(int.from_bytes(hashlib.sha256(b'nbls-bench-sk' + i.to_bytes(4, 'big')).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(lo, hi)]
            msgs_l = [hashlib.sha256(b'msg' + i.to_bytes(4, 'big')).digest() for i in range(lo, hi)]
            pks_l = eng.get_public_keys(sks_l)
            aff_l, _ = eng.sign_batch_affine(msgs_l, sks_l)
            psum, _ = eng.point_sum(aff_l, g2=True)                      # this rank's share of the aggregate signature (affine, 192 B)
            d_ps = torch.frombuffer(bytearray(psum), dtype=torch.uint8).cuda()
            if multi:
                allps = par.all_gather_bytes(d_ps)
                agg_sig, _ = eng.point_sum(bytes(allps.cpu().numpy().tobytes()), g2=True)
            else:
                agg_sig = psum
            sig = eng.compress_g2(agg_sig)
            uni_l = b''.join(oracle.expand_message_xmd(m, oracle_py.DST_DEFAULT, 256) for m in msgs_l)
            d_sig = torch.frombuffer(bytearray(sig), dtype=torch.uint8).cuda()
            d_uni = torch.frombuffer(bytearray(uni_l), dtype=torch.uint8).cuda()
            d_pk = torch.frombuffer(bytearray(b''.join(pks_l)), dtype=torch.uint8).cuda()
            be = par.EngineBackend(eng)
            assert par.verify_batch_sharded(be, d_sig, d_uni, d_pk) is True, 'sharded verifyBatch parity (true case) failed'
            if rank == 0 and lo == 0:      # spot check of the setup itself: the first key and signature share against the oracle
                assert pks_l[0] == oracle.get_public_key(sks_l[0])
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            s0 = time.perf_counter()
            sreps = 3
            for _ in range(sreps):
                par.verify_batch_sharded(be, d_sig, d_uni, d_pk)
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            sdt_ = time.perf_counter() - s0
            if multi:
                t = torch.tensor([sdt_], dtype=torch.float64, device='cuda')
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sdt_ = float(t.item())
            vshard = {'metric': 'verifyBatch sigs/sec', 'n_signatures': nv, 'value': round(nv * sreps / sdt_, 2), 'unit': 'sigs/s', 'ms': round(sdt_ / sreps * 1e3, 3),
                      'note': 'distinct 32-byte messages, 48-byte keys, one 96-byte aggregate signature; (key, message) pairs sharded over %d rank(s): decompress + hash-to-G2 + Miller product per rank, all-gather of 576 B Fp12 partials, shared final exponentiation; inputs (incl. expand_message_xmd output) resident in HBM' % world}
            del d_uni, d_pk
    except Exception as e:   # noqa: BLE001
        if multi:
            raise     # see the product leg: collectives inside
        vshard = {'error': repr(e)}

    # ---- BASELINE configs[3] as written (all N): --config3-pairings independent pairings sharded contiguously over the ranks, every rank ONE nbls_pairing_batch_dev call on its
    # shard of SURVEY 8(d)'s item stream, barrier before and after, whole-node pairings/s over the slowest rank; no collective on the data path
    config3 = None
    config3_local_ms = None
    try:
        if args.config3_pairings > 0:
            par = importlib.import_module('noble-bls12-381_amd.parallel')
            lo3, hi3 = par.shard_bounds(args.config3_pairings, world, rank)
            m3 = hi3 - lo3
            C1, C2 = synth_stream(eng, oracle, m3, seed=0x6e626c73 + 3, first=lo3, check=(0, -1))
            c1 = torch.frombuffer(bytearray(C1), dtype=torch.uint8).cuda(); c2 = torch.frombuffer(bytearray(C2), dtype=torch.uint8).cuda()
            co = torch.empty(576 * m3, dtype=torch.uint8, device='cuda')
            eng.pairing_batch_dev(m3, c1.data_ptr(), c2.data_ptr(), co.data_ptr(), True, stream)     # warm-up (scratch grows to this size)
            torch.cuda.synchronize()
            ref3, _ = oracle.pairing_batch(C1[:96 * 4] + C1[-96 * 4:], C2[:192 * 4] + C2[-192 * 4:], True, False, threads=8)
            assert bytes(co[:576 * 4].cpu().numpy().tobytes()) + bytes(co[-576 * 4:].cpu().numpy().tobytes()) == ref3, 'configs[3] parity check failed'
            # size-independent property over the whole shard: bilinearity of a random-looking pair of items is covered by tests; here a checksum of the outputs must not change between calls
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            q0 = time.perf_counter()
            creps = 2
            for _ in range(creps):
                eng.pairing_batch_dev(m3, c1.data_ptr(), c2.data_ptr(), co.data_ptr(), True, stream)
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            qdt = (time.perf_counter() - q0) / creps
            config3_local_ms = qdt * 1e3
            if multi:
                t = torch.tensor([qdt], dtype=torch.float64, device='cuda')
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                qdt = float(t.item())
            config3 = {'metric': 'pairings/sec, BASELINE configs[3]: %d independent pairings sharded over %d rank(s), one call per rank' % (args.config3_pairings, world),
                       'pairings': args.config3_pairings, 'pairings_per_rank': m3, 'value': round(args.config3_pairings / qdt, 2), 'unit': 'pairings/s', 'ms': round(qdt * 1e3, 3),
                       'roofline_frac_per_gpu': round(args.config3_pairings / world / qdt * (FPMUL_MILLER + FPMUL_FINALEXP) * MAD_PER_FPMUL / 1e12 / PEAK_TMAD, 4),
                       'note': 'inputs: items [lo, hi) of the SURVEY 8(d) stream per rank, resident in HBM; outputs 576 B per pairing into HBM; time = max over ranks between two barriers'}
            del c1, c2, co
    except Exception as e:   # noqa: BLE001
        if multi:
            raise
        config3 = {'error': repr(e)}

    # ---- one PROCESS over every GPU of the node (rank 0 of an N > 1 job, the other ranks wait at the barrier): nbls_multi_pairing_batch / nbls_multi_verify_batch from host
    # buffers -- the in-library form of the same sharding (csrc/nbls_multi.cpp: a context, a persistent host thread and a stream per device, partials gathered with hipMemcpyPeer)
    multi1 = None
    if world > 1:
        try:
            if rank == 0 and torch.cuda.device_count() >= world:
                me = pkg.MultiEngine(list(range(world)))
                nm1 = 16384 * world
                M1, M2 = synth_stream(eng, oracle, nm1, seed=0x6e626c73 + 4, check=(0,))
                outm, _ = me.pairing_batch(M1, M2, True, False)
                refm, _ = oracle.pairing_batch(M1[:96 * 2] + M1[-96 * 2:], M2[:192 * 2] + M2[-192 * 2:], True, False, threads=4)
                assert outm[:576 * 2] + outm[-576 * 2:] == refm, 'multi-device pairing parity check failed'
                u0 = time.perf_counter(); me.pairing_batch(M1, M2, True, False); udt = time.perf_counter() - u0
                prod, _ = me.miller_product(M1[:96 * 4096], M2[:192 * 4096], True, False)
                assert prod == eng.miller_product(M1[:96 * 4096], M2[:192 * 4096], True, False)[0], 'multi-device product differs from one device'
                multi1 = {'devices': world, 'pairings': nm1, 'pairings_per_s_host_buffers': round(nm1 / udt, 2), 'ms': round(udt * 1e3, 3),
                          'peer_access': [me.lib.nbls_multi_peer_access(me.h, i) for i in range(world)],
                          'note': 'nbls_multi_pairing_batch from HOST buffers (PCIe in and out included) in ONE process over all devices; product of 4096 pairs across the devices equals the one-device product'}
                me.close()
        except Exception as e:   # noqa: BLE001  -- no collective inside: rank 0 reaches the barrier either way
            multi1 = {'error': repr(e)}
        dist.barrier()

    # ---- what an N-rank record needs to explain itself (round-5 review item 5): every rank's own device, its own times of the value region and of its configs[3] shard, how many
    # ranks RCCL saw, and the peer-access matrix of the node as rank 0 sees it
    multi_gpu = None
    if multi:
        mine = {'rank': rank, 'local_rank': local_rank, 'device': torch.cuda.get_device_name(local_rank), 'value_region_ms': round(local_times[0] * 1e3, 3) if local_times else None,
                'config3_shard_ms': round(config3_local_ms, 3) if config3_local_ms is not None else None, 'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES')}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        ndev = torch.cuda.device_count()
        multi_gpu = {'per_rank': per_rank, 'rccl_ranks': ranks_seen if args.dist_backend == 'nccl' else None, 'backend': args.dist_backend, 'visible_devices': ndev,
                     'peer_access': [[int(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(ndev)] for i in range(ndev)] if rank == 0 else None,
                     'note': 'value / config3 divide by the SLOWEST rank (max-reduced between barriers); per_rank shows every rank\'s own time'}

    # ---- roofline leg: per-kernel HIP-event durations of the same step (separate untimed passes)
    roof = None
    cpu = None
    facade = None
    if rank == 0:
        eng.timing_enable(True)
        reps = 5
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        tm = eng.timing_read()
        eng.timing_enable(False)
        per_step = {k: v[0] / reps for k, v in tm.items()}          # ms per bench step, summed over that program's launches
        MILLER_PROGS = ('miller_fe', 'miller_fe_ls', 'miller_fe_ls2', 'lines_pq', 'acc_fe')   # one fused program, or LINES_PQ -> ACC_FE through line tables in HBM (which one ran is read off the timing slots below)
        ms_miller = sum(v for k, v in per_step.items() if k in MILLER_PROGS)
        ms_inv = per_step['fp_inv']
        ms_hard = sum(v for k, v in per_step.items() if k not in MILLER_PROGS + ('fp_inv',))
        mads = n * (FPMUL_MILLER + FPMUL_FINALEXP) * MAD_PER_FPMUL
        vm_ms = ms_miller + ms_hard
        call_ms = vm_ms + ms_inv            # every kernel of the call, the inversion kernel included (round 4 left it out of the denominator while naming it in the label)
        achieved = mads / (call_ms * 1e-3) / 1e12
        split_miller = per_step.get('lines_pq', 0) > 0      # the form that RAN (round 4 tested n >= 49152, a threshold the library no longer has)
        chained = per_step.get('fe_mid1', 0) == 0           # the final exponentiation's middle as one chained launch: its time is booked on `expx`
        # algorithmic HBM bytes per pairing: wire points in (288) and Fp12 out (576) + the raw scratch elements (768 B per Fp12, 64 B per Fp)
        # every phase program reads and writes: Miller W 832 | fp_inv R 64 W 64 | fe_easy R 832 W 768 | 5 x expx R 768 W 768 | fe_mid1/2 R 1536 W 768 | fe_final R 5376
        # -- and, with the two-program Miller loop, one 26,112-byte line table written by LINES_PQ and read by ACC_FE
        hbm_bytes = n * (288 + 832 + 128 + 832 + 768 + 5 * 1536 + 2 * 2304 + 5376 + 576 + (2 * 26112 if split_miller else 0))
        traffic = None; valu_busy = None
        try:   # HBM bytes measured with rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, see profiles/README.md), same batch size only
            with open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')) as fh:
                prof = json.load(fh).get(str(n), {})
                traffic = prof.get('bytes_per_step'); valu_busy = prof.get('valu_issue_busy')
        except OSError:
            pass
        bind = eng.kernel_bindings()
        ran = sorted({bind[k] for k in per_step if k in bind and per_step[k] > 0} | {'nbls_fp_inv_kernel'})
        n_aot = sum(1 for k in bind.values() if k.startswith('nbls_aot_'))
        roof = {
            'bound': 'valu-int32-mad', 'kernel': ' + '.join(ran), 'kernel_source': 'nbls_program_kernel() of the programs this call launched: the kernels that RAN (an interpreter binding would show as nbls_vm_kernel)',
            'aot_programs': '%d/%d' % (n_aot, len(bind)),     # step programs bound to an ahead-of-time kernel in this process (tests/test_gpu_binding.py asserts the same on the box)
            'miller_form': 'LINES_PQ -> ACC_FE (line tables through HBM)' if split_miller else 'one fused program', 'final_exp_middle': 'one chained launch' if chained else 'seven launches',
            'achieved': round(achieved, 4), 'peak': round(PEAK_TMAD, 3), 'unit': 'TMAD32/s', 'frac': round(achieved / PEAK_TMAD, 4),
            'traffic': traffic, 'traffic_source': 'profiles/hbm_traffic.json (rocprofv3 PMC passes of this batch size, FETCH_SIZE x 2 + WRITE_SIZE; a committed measurement, not taken in this run)' if traffic is not None else None,
            'valu_issue_busy': valu_busy,    # SQ_INSTS_VALU x 4 clocks / (kernel time x 2.4 GHz x 1024 SIMDs) from the PMC pass in profiles/ (same batch size, one batch at a time)
            'frac_at_value': round(value / world * (FPMUL_MILLER + FPMUL_FINALEXP) * MAD_PER_FPMUL / 1e12 / PEAK_TMAD, 4),
            'frac_note': 'achieved/frac: the kernels of ONE %d-pairing call running alone (algorithmic multiply-adds over the sum of the HIP-event durations of ALL its launches, the inversion kernel included; profiles/ holds the rocprofv3 kernel trace of the same command); frac_at_value: the same algorithmic work at the rate of `value` (%d calls overlapping on %d streams; profiles/ holds a kernel trace taken with the same --inflight)' % (n, D, D),
            'clock_note': 'peak is priced at 2.4 GHz; under saturated load (value, large_batch) this engine is power-limited: 2.21-2.22 GHz at 1330-1360 W of the 1400 W package limit (profiles/round4_clocks_under_load.txt), and a pure stream of its multiply-add sustains 30.6 T/s at 2.32 GHz (profiles/round4_ubench_mad_power.txt); one call at a time runs at 2.39 GHz',
            'kernel_ms': {k: round(v, 4) for k, v in per_step.items()},
            'miller_frac': round(n * FPMUL_MILLER * MAD_PER_FPMUL / (ms_miller * 1e-3) / 1e12 / PEAK_TMAD, 4) if ms_miller > 0 else None,      # (None: a batch size whose Miller program is not in MILLER_PROGS)
            'final_exp_frac': round(n * FPMUL_FINALEXP * MAD_PER_FPMUL / (ms_hard * 1e-3) / 1e12 / PEAK_TMAD, 4) if ms_hard > 0 else None,
            'hbm': {'algorithmic_bytes_per_launch': hbm_bytes, 'achieved_GBps': round(hbm_bytes / (call_ms * 1e-3) / 1e9, 3),
                    'peak_GBps': HBM_PEAK_GBPS, 'frac': round(hbm_bytes / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                    'algorithmic_bytes_8d': 864 * n, 'traffic_ratio_vs_8d': round(traffic / (864.0 * n), 1) if traffic else None,
                    'traffic_GBps': round(traffic / (call_ms * 1e-3) / 1e9, 1) if traffic else None,
                    'note': 'algorithmic_bytes_per_launch: wire I/O + the raw scratch between the launches of the form that ran (line tables included when the Miller loop ran as two programs); algorithmic_bytes_8d: SURVEY 8(d)\'s 864 B per pairing (points in, Fp12 out); traffic_ratio_vs_8d: the PMC-measured bytes (roofline.traffic) over that -- the phase scratch and the line tables, never the bound (frac)'},
        }
        # the same batch through the host-buffer entry point (PCIe in and out included) -- reported, never `value`
        h0 = time.perf_counter()
        for _ in range(3):
            eng.pairing_batch(G1, G2, True, False)
        roof['host_call_pairings_per_s'] = round(3 * n / (time.perf_counter() - h0), 2)
        # the same pipeline as ONE call of --large-batch pairings: every SIMD holds three wavefronts (the VGPR limit), the issue slots are saturated
        if args.large_batch > 0:
            nl = args.large_batch
            GL1, GL2 = synth_stream(eng, oracle, nl, seed=0x6e626c73, first=rank * n, check=(-1,))      # the same stream, nl items of it: every pair distinct
            dl1 = torch.frombuffer(bytearray(GL1), dtype=torch.uint8).cuda(); dl2 = torch.frombuffer(bytearray(GL2), dtype=torch.uint8).cuda()
            dlo = torch.empty(576 * nl, dtype=torch.uint8, device='cuda')
            for _ in range(2):
                eng.pairing_batch_dev(nl, dl1.data_ptr(), dl2.data_ptr(), dlo.data_ptr(), True, stream)
            torch.cuda.synchronize()
            ref_tail, _ = oracle.pairing_batch(GL1[-96 * 8:], GL2[-192 * 8:], True, False, threads=8)
            assert bytes(dlo[:576 * 8].cpu().numpy().tobytes()) == ref and bytes(dlo[-576 * 8:].cpu().numpy().tobytes()) == ref_tail, 'large-batch parity check failed'
            lreps = max(3, int(1.0 / (nl / 2.0e6)))      # about one second
            l0 = time.perf_counter()
            for _ in range(lreps):
                eng.pairing_batch_dev(nl, dl1.data_ptr(), dl2.data_ptr(), dlo.data_ptr(), True, stream)
            torch.cuda.synchronize()
            ldt = (time.perf_counter() - l0) / lreps
            eng.timing_enable(True)
            eng.pairing_batch_dev(nl, dl1.data_ptr(), dl2.data_ptr(), dlo.data_ptr(), True, stream); torch.cuda.synchronize()
            ltm = eng.timing_read(); eng.timing_enable(False)
            # the same call with three in flight (three contexts / streams, outputs of their own): the launch tails of one call are filled by the others
            LF = min(3, D)
            louts = [dlo] + [torch.empty(576 * nl, dtype=torch.uint8, device='cuda') for _ in range(LF - 1)]
            for i in range(LF):
                pipe.engines[i].pairing_batch_dev(nl, dl1.data_ptr(), dl2.data_ptr(), louts[i].data_ptr(), True, None)
            torch.cuda.synchronize()
            # shader clock and package power while this leg keeps the chip saturated (rocm-smi from a side thread, two samples; None when the tool is missing): the saturated legs are
            # power-limited, and the roofline peak is priced at 2.4 GHz
            clk = {'sclk_mhz': [], 'power_w': []}
            def _sample_clocks():
                import re as _re, shutil as _sh, subprocess as _sp
                if not _sh.which('rocm-smi'):
                    return
                for _ in range(2):
                    time.sleep(0.25)
                    try:
                        o = _sp.run(['rocm-smi', '-d', str(local_rank), '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10).stdout
                    except Exception:   # noqa: BLE001
                        return
                    m = _re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', o); w = _re.search(r'Power \(W\): ([0-9.]+)', o)
                    if m: clk['sclk_mhz'].append(int(m.group(1)))
                    if w: clk['power_w'].append(float(w.group(1)))
            import threading as _th
            smp = _th.Thread(target=_sample_clocks); smp.start()
            f0 = time.perf_counter()
            for k in range(LF * lreps):
                pipe.engines[k % LF].pairing_batch_dev(nl, dl1.data_ptr(), dl2.data_ptr(), louts[k % LF].data_ptr(), True, None)
            torch.cuda.synchronize()
            ldt_f = (time.perf_counter() - f0) / (LF * lreps)
            smp.join()
            assert bytes(louts[-1][:576 * 8].cpu().numpy().tobytes()) == ref, 'large-batch in-flight parity check failed'
            del louts
            # the call runs as two halves on two streams from 8192 pairs (csrc/pipelines_pairing.cpp): its kernels overlap, so achieved / frac are over the WALL time of the call
            roof['large_batch'] = {'pairings': nl, 'pairings_per_s': round(nl / ldt, 2), 'ms_per_call': round(ldt * 1e3, 3),
                                   'achieved': round(nl * (FPMUL_MILLER + FPMUL_FINALEXP) * MAD_PER_FPMUL / ldt / 1e12, 4),
                                   'frac': round(nl * (FPMUL_MILLER + FPMUL_FINALEXP) * MAD_PER_FPMUL / ldt / 1e12 / PEAK_TMAD, 4),
                                   'kernel_ms': {k: round(v[0], 4) for k, v in ltm.items()},
                                   'in_flight': {'calls_in_flight': LF, 'pairings_per_s': round(nl / ldt_f, 2), 'ms_per_call_amortised': round(ldt_f * 1e3, 3),
                                                 'roofline_frac': round(nl * (FPMUL_MILLER + FPMUL_FINALEXP) * MAD_PER_FPMUL / ldt_f / 1e12 / PEAK_TMAD, 4),
                                                 'sclk_mhz_under_load': clk['sclk_mhz'] or None, 'package_power_w_under_load': clk['power_w'] or None},
                                   'note': 'ONE call (the library runs it as two halves on two streams); achieved/frac over the wall time of the call; kernel_ms = HIP-event durations of its launches, which overlap pairwise (Miller loop as LINES + ACC)'}
            del dl1, dl2, dlo
        if world == 1 and not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            threads = min(cores, 256)
            sample = 16384 if threads >= 128 else 4096 if threads >= 16 else 512
            g1s, g2s = (G1 * (sample // n + 1))[:96 * sample], (G2 * (sample // n + 1))[:192 * sample]
            # every hardware thread, and -- because two threads of a core share its multiplier -- every other one: the better figure is reported
            best = None
            for th_ in sorted({threads, max(1, threads // 2)}, reverse=True):
                oracle.pairing_batch(g1s[:96 * th_], g2s[:192 * th_], True, False, threads=th_)   # warm
                c0 = time.perf_counter()
                oracle.pairing_batch(g1s, g2s, True, False, threads=th_)
                d_ = time.perf_counter() - c0
                if best is None or d_ < best[0]:
                    best = (d_, th_)
            cdt, threads_used = best
            all_threads = threads
            threads = threads_used
            c1 = time.perf_counter()
            oracle.pairing_batch(g1s[:96 * 64], g2s[:192 * 64], True, False, threads=1)
            cdt1 = time.perf_counter() - c1
            # the same algorithm in the reference's own language and number representation: JavaScript BigInt on one core (oracle/js_bigint_pairing.js;
            # the reference itself cannot travel to this box -- BASELINE.md section 4 holds its measured ratio to the reference under the same Node)
            jsb = None
            try:
                import shutil, subprocess
                if shutil.which('node'):
                    js = os.path.join(ROOT, 'oracle', 'js_bigint_pairing.js')
                    jr = json.loads(subprocess.run(['node', js, '5'], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
                    # the same on every host core at once: N single-threaded processes (the reference is single-threaded; a user with N cores runs N of them)
                    procs = [subprocess.Popen(['node', js, '5'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
                    outs = [p.communicate(timeout=300)[0] for p in procs]
                    rs = [json.loads(o.strip().splitlines()[-1]) for o in outs if o.strip()]
                    try:
                        ratio = json.load(open(os.path.join(ROOT, 'profiles', 'reference_ratio.json')))
                    except Exception:   # noqa: BLE001
                        ratio = None
                    rr = ratio['ratio'] if ratio else None
                    jsb = {'value': jr['pairings_per_s'], 'unit': 'pairings/s', 'cores': 1, 'kind': 'port', 'language': 'JavaScript BigInt (node %s)' % jr['node'], 'sample': '%d pairings in %.1f s, single thread' % (jr['pairings'], jr['seconds']),
                           'checked_against_reference_vector': bool(jr['ok']),
                           'all_cores': {'value': round(sum(r['pairings_per_s'] for r in rs), 2), 'unit': 'pairings/s', 'cores': len(rs), 'sample': '%d single-threaded processes side by side, %d pairings in %.1f s each on average' % (len(rs), sum(r['pairings'] for r in rs) // max(1, len(rs)), sum(r['seconds'] for r in rs) / max(1, len(rs))),
                                         'all_checked': all(r['ok'] for r in rs)},
                           'ratio_to_reference': rr, 'ratio_source': 'profiles/reference_ratio.json (tools/measure_reference_ratio.py in the build container, %s): this code / the real reference under the same Node' % (ratio['date'] if ratio else 'missing'),
                           'reference_estimate': {'one_core': round(jr['pairings_per_s'] / rr, 2), 'all_cores': round(sum(r['pairings_per_s'] for r in rs) / rr, 2), 'note': 'derived: measured here / committed ratio; the reference itself cannot travel to this box'} if rr else None,
                           'note': 'the reference algorithm over the facade\'s BigInt field classes (oracle/js_bigint_pairing.js)'}
            except Exception as e:   # noqa: BLE001
                jsb = {'error': repr(e)}
            cpu = {'value': round(sample / cdt, 2), 'unit': 'pairings/s', 'cores': threads, 'kind': 'port', 'js_bigint': jsb,
                   'sample': '%d pairings of the same workload on %d host threads (oracle/ C restatement; timed on %d and on %d threads, the faster is reported); 1 thread: %.1f pairings/s' % (sample, threads, all_threads, max(1, all_threads // 2), 64 / cdt1),
                   'host_cpu_count': cores,
                   'reference_figure': {'value': 38.6, 'unit': 'pairings/s per core', 'source': 'BASELINE.md section 2: the reference itself (noble-bls12-381 v1.4.0, TypeScript / bigint) under Node 12 in the build container; it does not travel to the GPU box, so the port above is what is timed here'}}
        # the JS facade from JavaScript: await bls.verify(...) / await bls.sign(...) as a user of the reference would call them (tools/bench_facade.js)
        facade = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                import shutil, subprocess
                addon = os.path.join(ROOT, 'noble-bls12-381_amd', 'js', 'nbls_napi.node')
                if shutil.which('node') and os.path.exists(addon):
                    facade = json.loads(subprocess.run(['node', os.path.join(ROOT, 'tools', 'bench_facade.js')], capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1])
            except Exception as e:   # noqa: BLE001
                facade = {'error': repr(e)}
        vbatch = None
        if world == 1 and args.verify_batch > 0:
            nv = args.verify_batch
            cores = os.cpu_count() or 1
            th = min(cores, 256)
            sks = [(int.from_bytes(hashlib.sha256(b'nbls-bench-sk' + i.to_bytes(4, 'big')).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(nv)]
            msgs = [hashlib.sha256(b'msg' + i.to_bytes(4, 'big')).digest() for i in range(nv)]
            # inputs made on the GPU (getPublicKey + sign ladders, point sum, compression: ~0.1 s instead of ~20 s of host threads), checked against the oracle on a sample:
            # the keys and the first signatures must be the oracle's, and the aggregate over the sample must verify on the CPU
            pks = eng.get_public_keys(sks)
            aff, _ = eng.sign_batch_affine(msgs, sks)
            sig = eng.compress_g2(eng.point_sum(aff, g2=True)[0])
            ns_chk = min(nv, 32)
            # a strided sample across the whole range, the last index included (round 4 looked at the first 32 only): keys and signatures must be the oracle's
            for i in sorted(set(list(range(0, nv, max(1, nv // 12))) + [nv - 1])):
                assert pks[i] == oracle.get_public_key(sks[i]) and eng.compress_g2(aff[192 * i:192 * i + 192]) == oracle.sign(msgs[i], sks[i])[1], 'verifyBatch input check against the oracle failed'
            assert oracle.verify_batch_mt(eng.compress_g2(eng.point_sum(aff[:192 * ns_chk], g2=True)[0]), msgs[:ns_chk], pks[:ns_chk], threads=min(th, 32)), 'verifyBatch sample does not verify on the CPU'
            assert eng.verify_batch(sig, msgs, pks) is True, 'verifyBatch parity (true case) failed'
            bad = list(msgs); bad[nv // 2] = bytes([bad[nv // 2][0] ^ 1]) + bad[nv // 2][1:]
            assert eng.verify_batch(sig, bad, pks) is False, 'verifyBatch parity (false case) failed'
            vreps = 3
            v0 = time.perf_counter()
            for _ in range(vreps):
                eng.verify_batch(sig, msgs, pks)
            vdt = (time.perf_counter() - v0) / vreps
            # everything resident in HBM: signature, message bytes + offsets, compressed keys; SHA-256 expand_message_xmd is part of the timed call
            import numpy as np
            d_sig = torch.frombuffer(bytearray(sig), dtype=torch.uint8).cuda()
            d_msg = torch.frombuffer(bytearray(b''.join(msgs)), dtype=torch.uint8).cuda()
            d_off = torch.from_numpy(np.cumsum([0] + [len(m) for m in msgs], dtype=np.uint32).astype(np.int32)).cuda()
            d_pk = torch.frombuffer(bytearray(b''.join(pks)), dtype=torch.uint8).cuda()
            vcall = lambda e: e.verify_batch_msgs_dev(nv, d_sig.data_ptr(), d_msg.data_ptr(), d_off.data_ptr(), d_pk.data_ptr())
            assert vcall(eng) is True
            torch.cuda.synchronize()
            dreps = 8      # mean of eight calls, one at a time (three left the figure at the mercy of one slow call: +-0.4 ms)
            v1 = time.perf_counter()
            for _ in range(dreps):
                vcall(eng)
            vdt_dev = (time.perf_counter() - v1) / dreps
            # two calls in flight (two engine contexts, two host threads; ctypes drops the GIL inside the call): the single-item tail of one call (product
            # tree + one final exponentiation, ~2.3 ms of pure latency) runs under the bulk of the next -- the sustained rate of a service verifying batch after batch
            import threading
            VF = max(2, args.verify_inflight)
            vengs = [eng] + [pipe.engines[i] if i < D else pkg.Engine(local_rank) for i in range(1, VF)]
            for e in vengs[1:]:
                e.set_ls_max(1024, 2048); e.set_inv_wide_max(4096)      # in-flight contexts of the pairing leg: the library's defaults for the one-item tail of a verifyBatch call
                assert vcall(e) is True
            preps = max(2, vreps)
            def _loop(e):
                for _ in range(preps):
                    vcall(e)
            torch.cuda.synchronize()
            ths = [threading.Thread(target=_loop, args=(e,)) for e in vengs]
            p1 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            vdt_pipe = (time.perf_counter() - p1) / (VF * preps)
            # latency of ONE verify (the reference's verify, index.ts:756-767: decode key and signature, hash the message, 2 Miller loops, 1 final exponentiation)
            sig1 = eng.compress_g2(aff[:192])
            assert eng.verify_batch(sig1, msgs[:1], pks[:1]) is True
            l0 = time.perf_counter()
            for _ in range(5):
                eng.verify_batch(sig1, msgs[:1], pks[:1])
            single_ms = (time.perf_counter() - l0) / 5 * 1e3
            ns = min(nv, 2048)
            sig_s = eng.compress_g2(eng.point_sum(aff[:192 * ns], g2=True)[0])
            c0 = time.perf_counter()
            okc = oracle.verify_batch_mt(sig_s, msgs[:ns], pks[:ns], threads=min(th, 64))
            cdt = time.perf_counter() - c0
            FPMUL_VERIFY = 13294 + 12452 + 610       # SURVEY.md 8(d), per signature: validity + Miller loop + product term; hash-to-G2; key decompression
            v_ach = nv * FPMUL_VERIFY * MAD_PER_FPMUL / vdt_dev / 1e12
            FPMUL_VERIFY_EXEC = FPMUL_VERIFY - 3351      # the reference re-checks the subgroup of every H(m) (PointG2.assertValidity, 3,351 Fp multiplications); the engine does not (a hash output is in the subgroup by construction)
            vbatch = {'metric': 'verifyBatch sigs/sec', 'n_signatures': nv, 'value': round(nv / vdt_dev, 2), 'unit': 'sigs/s',
                      'roofline': {'bound': 'valu-int32-mad', 'achieved': round(v_ach, 4), 'peak': round(PEAK_TMAD, 3), 'unit': 'TMAD32/s', 'frac': round(v_ach / PEAK_TMAD, 4),
                                   'frac_executed': round(nv * FPMUL_VERIFY_EXEC * MAD_PER_FPMUL / vdt_dev / 1e12 / PEAK_TMAD, 4), 'fpmul_executed_algorithm': FPMUL_VERIFY_EXEC,
                                   'note': 'algorithmic work per signature %d Fp multiplications x %d MAD32 (SURVEY 8(d)) over the wall time of the call (all kernels of all three streams)' % (FPMUL_VERIFY, MAD_PER_FPMUL)},
                      'note': 'distinct 32-byte messages, 48-byte keys, one 96-byte aggregate signature; SHA-256 expand_message_xmd + decompress + hash-to-G2 + %d Miller loops + 1 final exp on the GPU in the timed call; inputs (signature, message bytes, keys) resident in HBM' % (nv + 1),
                      'ms': round(vdt_dev * 1e3, 3),
                      'in_flight': {'calls_in_flight': VF, 'sigs_per_s': round(nv / vdt_pipe, 2), 'ms_per_call_amortised': round(vdt_pipe * 1e3, 3), 'roofline_frac': round(nv * FPMUL_VERIFY * MAD_PER_FPMUL / vdt_pipe / 1e12 / PEAK_TMAD, 4),
                                        'note': '%d verifyBatch calls of %d signatures overlapping (one context and host thread each): the latency-bound tail of one call runs under the bulk of the others; `value` / `ms` above are ONE call at a time' % (VF, nv)},
                      'host_call_sigs_per_s': round(nv / vdt, 2), 'host_call_ms': round(vdt * 1e3, 3),
                      'single_verify_ms': round(single_ms, 3), 'host_call_note': 'full nbls_verify_batch from host buffers: PCIe copies of messages, keys and signature included; SHA-256 expand_message_xmd runs on the device',
                      'cpu_baseline': {'value': round(ns / cdt, 2), 'unit': 'sigs/s', 'cores': min(th, 64), 'kind': 'port', 'sample': '%d signatures (sign-side setup included in neither)' % ns, 'ok': int(okc)}}
        sleg = None
        if world == 1 and args.sign_batch > 0:
            ns_ = args.sign_batch
            sks_ = [(int.from_bytes(hashlib.sha256(b'nbls-bench-sk2' + i.to_bytes(4, 'big')).digest(), 'big') % (2 ** 254) + 1).to_bytes(32, 'big') for i in range(ns_)]
            msgs_ = [hashlib.sha256(b'msg2' + i.to_bytes(4, 'big')).digest() for i in range(ns_)]
            sg = eng.sign_batch(msgs_, sks_)
            for i in (0, 1, ns_ // 2, ns_ - 1):       # parity spot check against the oracle
                assert sg[i] == oracle.sign(msgs_[i], sks_[i])[1], 'sign parity check failed'
            # `value`: the C-ABI call from host buffers by itself -- inputs packed and outputs allocated beforehand (round 4 timed the Python marshalling with it, and ONE call with the per-kernel
            # instrumentation on: 5.9 ms where the call takes 4.9), mean of four calls; `resident`: nbls_sign_batch_dev, messages / keys / signatures in HBM (the verifyBatch leg's convention)
            import ctypes as C_
            import numpy as np
            blob_ = b''.join(msgs_); kblob_ = b''.join(sks_)
            offs_np = np.zeros(ns_ + 1, dtype=np.uint32); offs_np[1:] = np.cumsum([len(m) for m in msgs_])
            offs_ = (C_.c_uint32 * (ns_ + 1)).from_buffer(offs_np)
            out_ = C_.create_string_buffer(192 * ns_); st_ = C_.create_string_buffer(ns_)
            eng.sign_packed(ns_, blob_, offs_, kblob_, out_, st_)
            assert eng.compress_batch(out_.raw[:192 * 4], g2=True) == b''.join(sg[:4]) and st_.raw == bytes(ns_), 'sign (packed buffers) differs'
            sreps = 4
            s0 = time.perf_counter()
            for _ in range(sreps):
                eng.sign_packed(ns_, blob_, offs_, kblob_, out_, st_)
            sdt = (time.perf_counter() - s0) / sreps
            d_m = torch.frombuffer(bytearray(blob_), dtype=torch.uint8).cuda(); d_o = torch.from_numpy(offs_np.view(np.int32)).cuda(); d_k = torch.frombuffer(bytearray(kblob_), dtype=torch.uint8).cuda()
            d_so = torch.empty(192 * ns_, dtype=torch.uint8, device='cuda'); d_ss = torch.empty(ns_, dtype=torch.uint8, device='cuda')
            eng.sign_batch_dev(ns_, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), d_so.data_ptr(), d_ss.data_ptr())
            assert bytes(d_so[:192 * 4].cpu().numpy().tobytes()) == out_.raw[:192 * 4] and bytes(d_so[-192:].cpu().numpy().tobytes()) == out_.raw[-192:], 'sign (resident) differs from the host-buffer call'
            r0 = time.perf_counter()
            for _ in range(sreps):
                eng.sign_batch_dev(ns_, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), d_so.data_ptr(), d_ss.data_ptr())
            rdt = (time.perf_counter() - r0) / sreps
            eng.timing_enable(True)
            eng.sign_batch_dev(ns_, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), d_so.data_ptr(), d_ss.data_ptr())
            stm = eng.timing_read(); eng.timing_enable(False)
            eng.point_mul_batch(sks_[:64])     # program upload outside the timed call (the verifyBatch leg, when it runs, has done it already)
            k0 = time.perf_counter()
            for _ in range(sreps):
                eng.point_mul_batch(sks_)
            kdt = (time.perf_counter() - k0) / sreps
            c0 = time.perf_counter()
            for i in range(16):
                oracle.sign(msgs_[i], sks_[i])
            csdt = (time.perf_counter() - c0) / 16
            sleg = {'metric': 'sign sigs/sec: nbls_sign_batch from host buffers (device SHA-256 expand_message_xmd + hash-to-G2 + constant-time G2 ladder + affine), compression not included', 'n': ns_, 'value': round(ns_ / sdt, 2), 'ms': round(sdt * 1e3, 3),
                    'resident': {'sigs_per_s': round(ns_ / rdt, 2), 'ms': round(rdt * 1e3, 3), 'kernels_ms': round(sum(v_[0] for v_ in stm.values()), 3),
                                 'note': 'nbls_sign_batch_dev: message bytes, offsets, keys and the affine signatures resident in HBM (as the verify_batch leg); kernels_ms = sum of the HIP-event durations of its launches (separate instrumented call)'},
                    'g2_ladder_kernel_ms': round(sum(stm.get(k_, (0, 0))[0] for k_ in ('g2_mul', 'g2_mul_w3', 'g2_mul_gls', 'g2_mul_sac', 'g2_mul_sac_ls2')), 3), 'get_public_key_keys_per_s': round(ns_ / kdt, 2),
                    'cpu_baseline': {'value': round(1 / csdt, 2), 'unit': 'sigs/s', 'cores': 1, 'kind': 'port', 'sample': '16 signatures on one host thread (oracle/)'}}
        aleg = None
        if world == 1 and args.sign_batch > 0:
            # aggregatePublicKeys / aggregateSignatures (index.ts:771-788; benchmark.js sizes 8..2048) on affine points from host buffers
            na = 2048
            g1p = synth_points(oracle, 64, seed=5)[0]; g2p = synth_points(oracle, 64, seed=6)[1]
            P1 = (g1p * (na // 64))[:96 * na]; P2 = (g2p * (na // 64))[:192 * na]
            assert eng.point_sum(P1[:96 * 8])[0] == oracle.g1_sum(P1[:96 * 8])[1] and eng.point_sum(P2[:192 * 8], g2=True)[0] == oracle.g2_sum(P2[:192 * 8])[1], 'aggregate parity check failed'
            eng.point_sum(P1); eng.point_sum(P2, g2=True)
            a0 = time.perf_counter(); eng.point_sum(P1); a1 = time.perf_counter(); eng.point_sum(P2, g2=True); a2 = time.perf_counter()
            big = (g1p * 1024)[:96 * 65536]
            eng.point_sum(big); a3 = time.perf_counter(); eng.point_sum(big); a4 = time.perf_counter()
            hm = [hashlib.sha256(b'h2c' + i.to_bytes(4, 'big')).digest() for i in range(16384)]
            hv = eng.hash_to_g2_batch(hm)
            assert hv[:192] == oracle.hash_to_g2(hm[0])[1] and hv[-192:] == oracle.hash_to_g2(hm[-1])[1], 'hash-to-G2 parity check failed'
            h0 = time.perf_counter(); eng.hash_to_g2_batch(hm); h1 = time.perf_counter()
            hg1 = eng.hash_to_curve_batch(hm, g2=False)
            assert hg1[:96] == oracle.hash_to_g1(hm[0])[1], 'hash-to-G1 parity check failed'
            h2 = time.perf_counter(); eng.hash_to_curve_batch(hm, g2=False); h3 = time.perf_counter()
            aleg = {'hash_to_g2_msgs_per_s': round(16384 / (h1 - h0), 2), 'hash_to_g1_msgs_per_s': round(16384 / (h3 - h2), 2), 'hash_note': '16384 32-byte messages from host buffers -> affine points (SHA-256 expand_message_xmd, SWU, isogeny, cofactor clearing on the GPU)',
                    'aggregate_public_keys_2048_ms': round((a1 - a0) * 1e3, 3), 'aggregate_signatures_2048_ms': round((a2 - a1) * 1e3, 3),
                    'aggregate_public_keys_65536_ms': round((a4 - a3) * 1e3, 3), 'note': 'affine points in host memory -> one affine sum (tree of complete additions on the GPU)'}
        mleg = None
        if world == 1 and args.msm_points > 0:
            nm = args.msm_points
            gen1 = oracle.g1_generator()
            a64 = [int.from_bytes(hashlib.sha256(b'msm-a' + bytes([i])).digest(), 'big') % R_ORDER for i in range(64)]
            p64 = [oracle.g1_mul(gen1, a)[1] for a in a64]
            ks = [int.from_bytes(hashlib.sha256(b'msm-k' + i.to_bytes(4, 'big')).digest(), 'big') % R_ORDER for i in range(nm)]
            Pm = b''.join(p64[i % 64] for i in range(nm)); Km = b''.join(k.to_bytes(32, 'big') for k in ks)
            d_pm = torch.frombuffer(bytearray(Pm), dtype=torch.uint8).cuda(); d_km = torch.frombuffer(bytearray(Km), dtype=torch.uint8).cuda()
            d_om = torch.empty(96, dtype=torch.uint8, device='cuda'); d_sm = torch.empty(1, dtype=torch.int8, device='cuda')
            st_ = torch.cuda.current_stream().cuda_stream
            eng.msm_dev(False, nm, d_pm.data_ptr(), d_km.data_ptr(), 255, d_om.data_ptr(), d_sm.data_ptr(), st_); torch.cuda.synchronize()
            expect = oracle.g1_mul(gen1, sum(a64[i % 64] * k for i, k in enumerate(ks)) % R_ORDER)[1]      # parity: (sum a_i k_i) G with one oracle multiplication
            assert bytes(d_om.cpu().numpy().tobytes()) == expect, 'msm parity check failed'
            m0 = time.perf_counter()
            for _ in range(5):
                eng.msm_dev(False, nm, d_pm.data_ptr(), d_km.data_ptr(), 255, d_om.data_ptr(), d_sm.data_ptr(), st_)
            torch.cuda.synchronize(); mdt = (time.perf_counter() - m0) / 5
            h0 = time.perf_counter(); eng.msm(Pm, [Km[32 * i:32 * i + 32] for i in range(nm)]); hdt = time.perf_counter() - h0
            nc = 256
            c0 = time.perf_counter()
            oracle.g1_sum(b''.join(oracle.g1_mul(p64[i % 64], ks[i])[1] for i in range(nc)))
            cmdt = (time.perf_counter() - c0) / nc
            mleg = {'metric': 'G1 multi-scalar multiplication points/sec (255-bit scalars, bucket method)', 'n': nm, 'value': round(nm / mdt, 2), 'ms': round(mdt * 1e3, 3),
                    'host_call_points_per_s': round(nm / hdt, 2), 'note': 'value: points and scalars resident in HBM; host_call: nbls_g1_msm from host buffers incl. PCIe and Python marshalling',
                    'cpu_baseline': {'value': round(1 / cmdt, 2), 'unit': 'points/s', 'cores': 1, 'kind': 'port', 'sample': '%d scalar multiplications + sum on one host thread (oracle/ double-and-add, no bucket method)' % nc}}
        line = {
            'metric': 'pairings/sec', 'value': round(value, 2), 'unit': 'pairings/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'int64', 'dtype_detail': 'signed 64-bit column accumulators over 14 x 28-bit limbs (v_mad_i64_i32), 381-bit Fp in Montgomery form R = 2^392; results bit-exact', 'data': 'synthetic',
            'config': {'workload': 'batch of %d independent BLS12-381 pairings per GPU (Miller loop + final exponentiation, inputs pre-validated, bit-exact vs reference), inputs/outputs resident in HBM' % n,
                       'batch_per_gpu': n, 'sharding': 'independent batches per rank, no collective', 'batches_in_flight': D,
                       'protocol': '`value` = exactly W warm-up + K timed steps with no other untimed steps in front of them (the round 1-4 protocol); the figure after %d further untimed steps is `steady_state`' % PREWARM},
            # the scalars a reader of a truncated line needs, before the large nested objects (round-5 review item 4)
            'summary': {'value_pairings_per_s': round(value, 2), 'steady_state_pairings_per_s': round(n * args.steps * world / dt_steady, 2) if dt_steady else None, 'prewarm_steps_of_steady_state': PREWARM if dt_steady else 0,
                        'single_call_ms': round(dt_serial / args.steps * 1e3, 4), 'single_call_roofline_frac': roof['frac'] if roof else None,
                        'large_batch_ms': (roof or {}).get('large_batch', {}).get('ms_per_call') if roof else None,
                        'verify_batch_ms': (vbatch or {}).get('ms'), 'verify_batch_in_flight_ms': ((vbatch or {}).get('in_flight') or {}).get('ms_per_call_amortised'), 'single_verify_ms': (vbatch or {}).get('single_verify_ms'),
                        'facade_verify_ms': (facade or {}).get('verify_ms') if isinstance(facade, dict) else None, 'facade_sign_ms': (facade or {}).get('sign_ms') if isinstance(facade, dict) else None,
                        'sign_sigs_per_s': (sleg or {}).get('value'), 'msm_points_per_s': (mleg or {}).get('value'), 'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES')},
            'steady_state': {'pairings_per_s': round(n * args.steps * world / dt_steady, 2), 'ms_per_step': round(dt_steady / args.steps * 1e3, 4), 'prewarm_steps': PREWARM,
                             'note': 'the same W warm-up + K timed steps after %d further untimed steps (a chip that has been busy for ~0.1 s: no clock ramp inside the region); NOT `value`' % PREWARM} if dt_steady else None,
            'cold_start': {'pairings_per_s': round(value, 2), 'ms_per_step': round(dt / args.steps * 1e3, 4), 'note': 'alias of `value` (round-5 name): W warm-up + K timed steps with nothing in front of them'},
            'in_flight': {'pairings_per_s': round(value, 2), 'batches_in_flight': D, 'ms_per_batch_amortised': round(dt / args.steps * 1e3, 4), 'timed_s': round(dt, 3),
                          'roofline_frac': round(value / world * (FPMUL_MILLER + FPMUL_FINALEXP) * MAD_PER_FPMUL / 1e12 / PEAK_TMAD, 4),
                          'note': '`value`: %d independent calls of %d pairings overlapping on %d streams / engine contexts per GPU' % (D, n, D)},
            'single_call': {'pairings_per_s': round(n * args.steps / dt_serial, 2), 'ms_per_batch': round(dt_serial / args.steps * 1e3, 4), 'roofline_frac': roof['frac'] if roof else None,
                            'note': 'one %d-pairing call at a time on one stream (this rank): the latency of a call; its roofline is the top-level `roofline` object' % n},
            'single_stream': {'pairings_per_s': round(n * args.steps / dt_serial, 2), 'ms_per_batch': round(dt_serial / args.steps * 1e3, 4), 'note': 'alias of single_call (round-1 name)'},
            'rccl_ranks': ranks_seen if (multi and args.dist_backend == 'nccl') else None, 'ranks_in_all_gather': ranks_seen, 'dist_backend': args.dist_backend if multi else None,
            'roofline': roof, 'cpu_baseline': cpu, 'facade': facade, 'product': product, 'verify_batch': vbatch if vbatch is not None else vshard, 'verify_batch_sharded': vshard if vbatch is not None else None, 'sign': sleg, 'aggregate': aleg, 'msm': mleg,
            'config3': config3, 'multi_one_process': multi1, 'multi_gpu': multi_gpu,
            'pool': {'depth': D, 'entry_point': 'nbls_pool_pairing_batch_dev (include/nbls.h): `value` is measured through the C ABI pool; noble-bls12-381_amd/pipeline.py only forwards to it'},
        }
        out_line = json.dumps(line)
    else:
        out_line = None
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if out_line is not None:      # the ONE JSON line is the last thing on stdout (RCCL prints a version banner there on some builds)
        sys.stdout.flush()
        print(out_line, flush=True)


if __name__ == '__main__':
    main()
