"""Data-parallel surrogate training: ONE flat gradient bucket, ONE all-reduce per step.

The reference trains the surrogate in a single CPU process (pdecontrol/mbrl/mbrl.py:357-365 builds a
plain ``pl.Trainer``; there is no torch.distributed anywhere).  On an 8 x MI355X node the batch is
sharded by rank and the 9 739-parameter gradient (38 956 bytes) is summed with one RCCL all-reduce
over xGMI -- a latency-bound message, so it is sent as a single contiguous bucket rather than
per-parameter: every ``param.grad`` is a view into ``FlatGradBucket.flat``.
"""
import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        """Sum over ranks then divide by the world size (equal shard sizes -> global-batch mean)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
        return self.flat

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank ``src``'s weights (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()


def shard_batch(batch, rank, world_size):
    """Contiguous equal shards of the leading (batch) dimension."""
    def cut(t):
        per = t.size(0) // world_size
        return t[rank * per:(rank + 1) * per]
    return tuple(cut(t) for t in batch)
