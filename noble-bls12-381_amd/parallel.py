"""Multi-GPU sharding of the pairing path: one process per GPU.

* Independent pairings shard by contiguous slices with no collective (see bench.py).
* The multi-pairing product  prod_i millerLoop(P_i, Q_i)  (the core of verify / verifyBatch, reference index.ts:811-816)
  has ONE exchange step: every rank reduces its shard to a single Fp12 partial (576 wire bytes), the partials are
  all-gathered (RCCL over xGMI; Fp12 multiplication is not an NCCL reduction op, so "all-reduce" = all-gather + local
  product), and every rank finishes with the same product and one shared final exponentiation.  Fp12 multiplication is
  commutative and results are canonical, so the rank order cannot change a bit of the result.
"""
import os

import torch


def _collective(group):
    """True when the exchange step has to run: a process group with more than one rank -- or NBLS_FORCE_COLLECTIVES=1, which runs the RCCL calls even
    with a single rank (tests/test_gpu_rccl.py: the only way to execute the nccl code path on a one-GPU box)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get('NBLS_FORCE_COLLECTIVES') == '1'


def all_gather_bytes(part, group=None):
    """all-gather of one small uint8 tensor per rank into one tensor (rank order).  RCCL gathers device tensors directly; gloo (CPU tests, and GPU runs whose
    ranks share a device) has no all-gather for device tensors, so the bytes are staged through the host there."""
    import torch.distributed as dist
    w = dist.get_world_size(group)
    if part.is_cuda and dist.get_backend(group) == 'gloo':
        host = part.cpu()
        out = torch.empty(w * host.numel(), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(part.device)
    out = torch.empty(w * part.numel(), dtype=part.dtype, device=part.device)
    dist.all_gather_into_tensor(out, part, group=group)
    return out


class EngineBackend:
    """GPU backend: device-resident uint8 tensors, work enqueued on the current torch stream."""

    def __init__(self, engine):
        self.eng = engine

    def local_product(self, g1, g2):
        n = g1.numel() // 96
        out = torch.empty(576, dtype=torch.uint8, device=g1.device)
        self.eng.miller_product_dev(n, g1.data_ptr(), g2.data_ptr(), out.data_ptr(), final_exp=False, stream=torch.cuda.current_stream().cuda_stream)
        return out

    def verify_partial(self, sig96, uniform, pks48):
        """sig96: device tensor or None; uniform: n x 256 expand_message_xmd bytes; pks48: n x 48 compressed keys (device tensors)"""
        n = pks48.numel() // 48
        out = torch.zeros(576, dtype=torch.uint8, device=pks48.device)
        zero = self.eng.verify_batch_partial_dev(n, sig96.data_ptr() if sig96 is not None else None, uniform.data_ptr(), pks48.data_ptr(), out.data_ptr(),
                                                 stream=torch.cuda.current_stream().cuda_stream)
        return out, zero

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
    if _collective(group):
        gathered = all_gather_bytes(part, group)
    else:
        gathered = part
    return backend.finish(gathered, final_exp)


ONE_FP12 = bytes(47) + b'\x01' + bytes(528)


def verify_batch_sharded(backend, sig96, msgs_local, pks_local, group=None):
    """verifyBatch (reference index.ts:792-821) with the (key, message) pairs sharded over the ranks: every rank decodes and hashes
    its own shard and reduces it to one Fp12 partial; rank 0 also contributes millerLoop(-G, S).  ONE exchange: the all-gather of
    the 576-byte partials (plus one scalar all-reduce carrying the "zero point met" flag); every rank then multiplies the partials,
    runs the shared final exponentiation and compares with ONE.  msgs_local is whatever the backend's verify_partial takes
    (EngineBackend: n x 256 expand_message_xmd bytes on the device)."""
    import torch.distributed as dist
    multi = _collective(group)
    rank = dist.get_rank(group) if multi else 0
    err = None
    try:
        part, zero = backend.verify_partial(sig96 if rank == 0 else None, msgs_local, pks_local)
    except Exception as e:      # an undecodable key / signature on this rank (the reference throws): every rank has to learn of it, or the others would wait in the collective
        if not multi:
            raise
        err, zero, part = e, False, torch.zeros(576, dtype=torch.uint8, device=pks_local.device if hasattr(pks_local, 'device') else 'cpu')
    if multi:
        flag = torch.tensor([2 if err is not None else (1 if zero else 0)], dtype=torch.int32, device=part.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) == 2:
            raise err if err is not None else RuntimeError('verify_batch_sharded: an input failed to decode on another rank')
        zero = bool(flag.item())
        gathered = all_gather_bytes(part, group)
    else:
        gathered = part
    if zero:
        return False
    res = backend.finish(gathered, True)
    return bytes(res.cpu().numpy().tobytes()) == ONE_FP12
