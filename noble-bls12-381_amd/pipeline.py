"""Batches in flight: D engine contexts on one GPU, each with its own HIP stream and scratch, fed round-robin (a thin wrapper over the C-ABI pool, nbls_pool_*).

A call of 4096 pairings fills the chip exactly one wavefront deep (1024 workgroups on 1024 SIMDs), and a lone wavefront reaches well under half of a SIMD's
issue rate, so one batch at a time leaves most of the machine idle.  Consecutive batches are independent, so a service keeps several of
them in flight: the kernels of batch i+1 run beside those of batch i on other streams.  Round 5, 4096-pairing batches: 1.77 M pairings/s one call at a time
(2.32 ms each), 3.0-3.1 M with twelve in flight (more contexts add nothing; fewer than seven lose).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES
hardware queues (default 4); streams that share a queue serialise, so set GPU_MAX_HW_QUEUES to at least the number of streams -- and to no more than 22: from
~24 user queues a process oversubscribes the hardware queue slots and every launch on the surplus queues pays a queue switch (bench.py: 22;
tools/ab_queues20.sh) -- in the environment before the runtime initialises.  A short burst of 20 batches is as fast on ten to fourteen contexts as on twenty
(2.95-2.97 against 2.93 M), and any burst that starts on an idle chip loses 3-6 % to the clock ramp (tools/burst_ab.py, bench.py `cold_start`).
"""
import ctypes as C

from .engine import Engine, NblsError, load_library


class PairingPipeline:
    """Round 5: a thin view of the C ABI's pool (include/nbls.h nbls_pool_*: `depth` contexts, each with its own stream and scratch, tuned for overlapping calls --
    the two-program Miller loop at every size, the final exponentiation's middle as seven launches -- and fed round-robin); `engines` wraps the pool's contexts."""

    def __init__(self, device_id=0, depth=12):
        assert depth >= 1
        self.lib = load_library()
        h = C.c_void_p()
        r = self.lib.nbls_pool_init(device_id, depth, C.byref(h))
        if r != 0:
            raise NblsError('nbls_pool_init failed: %s (code %d)' % (self.lib.nbls_strerror(r).decode(), r))
        self.h = h
        self.depth = depth
        self.engines = [Engine(_handle=self.lib.nbls_pool_context(self.h, i)) for i in range(depth)]

    def close(self):
        if getattr(self, 'h', None):
            for e in self.engines:
                e.close()
            self.lib.nbls_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def slot(self):
        """index (0 .. depth-1) of the context the next submit() will use: callers keep one output buffer per slot"""
        return self.lib.nbls_pool_next_slot(self.h)

    def submit(self, n, d_g1, d_g2, d_out, with_final_exp=True):
        """enqueue pairing(P_i, Q_i) for n device-resident pairs on the next context's own stream and return at once; the
        caller gives every batch in flight its own output buffer (one per `slot`) and calls synchronize() before reading"""
        r = self.lib.nbls_pool_pairing_batch_dev(self.h, n, d_g1, d_g2, int(with_final_exp), d_out, None)
        if r != 0:
            raise NblsError('nbls_pool_pairing_batch_dev: %s (code %d)' % (self.lib.nbls_strerror(r).decode(), r))

    def synchronize(self):
        r = self.lib.nbls_pool_synchronize(self.h)
        if r != 0:
            raise NblsError('nbls_pool_synchronize: code %d' % r)
