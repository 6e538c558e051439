"""Batches in flight: D engine contexts on one GPU, each with its own HIP stream and scratch, fed round-robin.

A call of 4096 pairings fills the chip exactly one wavefront deep (1024 workgroups on 1024 SIMDs), and a lone wavefront
reaches well under half of a SIMD's issue rate, so one batch at a time leaves most of the machine
idle.  Consecutive batches are independent, so a service keeps several of them in flight: the kernels of batch i+1 run
beside those of batch i on other streams.  Throughput at 4096-pairing batches rises from 1.3 M (one call at a time, 3.1 ms each) to 2.5 M pairings/s with seven to twenty batches in
flight (bench.py; 2.15 M with four, 2.31 M with five, 2.42 M with six, 2.47 M with seven, 2.52 M with ten or twelve, 2.55 M with fourteen).  The HIP runtime
multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise (eight batches on eight queues: 2.22 M, on
sixteen queues: 2.50 M), so set GPU_MAX_HW_QUEUES to at least the number of streams -- and to no more than 22: from ~24 user queues a process oversubscribes the hardware queue slots and every launch on the
surplus queues pays a queue switch (bench.py: 22; tools/ab_queues20.sh) -- in the environment before the runtime initialises when more than three batches are
kept in flight.  Round 4: 3.0-3.1 M pairings/s with twelve batches in flight; streams that carry several batches each run phase-locked (like that many large batches one after the other), so a short
burst of k batches is fastest on k streams (20 batches: 2.86 M on twenty streams, 2.79 M on ten).
"""
import os

from .engine import Engine


class PairingPipeline:
    def __init__(self, device_id=0, depth=12):
        assert depth >= 1
        self.engines = [Engine(device_id) for _ in range(depth)]
        self.depth = depth
        # With several batches in flight the SIMDs are shared by wavefronts of different calls, so what counts is the instruction count per
        # pairing, not the length of one call's longest instruction stream: the two-program Miller loop (15 % fewer instructions) is used
        # whatever the batch size.  A single context keeps the library's latency-oriented default (one fused program below 8192 pairs).
        # For the same reason the final exponentiation's middle runs as seven launches rather than one chain: a chained wavefront is 427 k instructions long, and with other calls'
        # wavefronts on the SIMDs the finer launches pack better (twenty calls on twenty streams +1.2 %, 512 calls twelve deep +0.8 %; tools/ab_chain20.sh).
        if depth > 1:
            for e in self.engines:
                e.set_split_miller_min(0)
                if os.environ.get('NBLS_PIPELINE_CHAIN') != '1':     # A/B switch (tools/ab_pipeline.sh)
                    e.set_chain_max(0)
        self._next = 0

    @property
    def slot(self):
        """index (0 .. depth-1) of the context the next submit() will use: callers keep one output buffer per slot"""
        return self._next % self.depth

    def submit(self, n, d_g1, d_g2, d_out, with_final_exp=True):
        """enqueue pairing(P_i, Q_i) for n device-resident pairs on the next context's own stream and return at once; the
        caller gives every batch in flight its own output buffer (one per `slot`) and calls synchronize() before reading"""
        e = self.engines[self._next % self.depth]
        self._next += 1
        e.pairing_batch_dev(n, d_g1, d_g2, d_out, with_final_exp, None)

    def synchronize(self):
        self.engines[0].device_synchronize()
