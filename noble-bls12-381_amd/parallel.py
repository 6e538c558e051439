"""Multi-GPU sharding of the pairing path: one process per GPU.

* Independent pairings shard by contiguous slices with no collective (see bench.py).
* The multi-pairing product  prod_i millerLoop(P_i, Q_i)  (the core of verify / verifyBatch, reference index.ts:811-816)
  has ONE exchange step: every rank reduces its shard to a single Fp12 partial (576 wire bytes), the partials are
  all-gathered (RCCL over xGMI; Fp12 multiplication is not an NCCL reduction op, so "all-reduce" = all-gather + local
  product), and every rank finishes with the same product and one shared final exponentiation.  Fp12 multiplication is
  commutative and results are canonical, so the rank order cannot change a bit of the result.
"""
import torch


class EngineBackend:
    """GPU backend: device-resident uint8 tensors, work enqueued on the current torch stream."""

    def __init__(self, engine):
        self.eng = engine

    def local_product(self, g1, g2):
        n = g1.numel() // 96
        out = torch.empty(576, dtype=torch.uint8, device=g1.device)
        self.eng.miller_product_dev(n, g1.data_ptr(), g2.data_ptr(), out.data_ptr(), final_exp=False, stream=torch.cuda.current_stream().cuda_stream)
        return out

    def finish(self, partials, final_exp=True):
        w = partials.numel() // 576
        out = torch.empty(576, dtype=torch.uint8, device=partials.device)
        self.eng.fp12_product_final_dev(w, partials.data_ptr(), out.data_ptr(), final_exp=final_exp, stream=torch.cuda.current_stream().cuda_stream)
        return out


def shard_bounds(n, world, rank):
    """contiguous split [lo, hi) of n items over `world` ranks"""
    return n * rank // world, n * (rank + 1) // world


def miller_product_sharded(backend, g1_local, g2_local, group=None, final_exp=True):
    """g1_local / g2_local: this rank's shard (uint8 tensors, 96 / 192 bytes per pair).  Returns the 576-byte result
    (identical on every rank)."""
    import torch.distributed as dist
    part = backend.local_product(g1_local, g2_local)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        w = dist.get_world_size(group)
        gathered = torch.empty(w * 576, dtype=torch.uint8, device=part.device)
        dist.all_gather_into_tensor(gathered, part, group=group)
    else:
        gathered = part
    return backend.finish(gathered, final_exp)
