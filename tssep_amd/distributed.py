"""Data-parallel gradient exchange: one process per GPU, one flat fp32 bucket, one
all-reduce(SUM) over RCCL/xGMI per optimizer step.

The reference has no distributed path (tssep/train/experiment.py:181-184 refuses >1 GPU).  Its
loss is SUMMED over the batch (tssep/train/model.py:669), so gradients of the shards are summed,
not averaged, to equal a single-process run over the concatenated batch.  Utterances are
independent in forward and backward, so the data path needs no other collective.
"""
import torch
import torch.distributed as dist


class GradBucket:
    """Flat view over all parameter gradients: grads live inside ONE contiguous buffer, so the
    all-reduce needs no pack/unpack copies.

    ``replicas`` > 1 keeps one extra flat buffer per concurrently running micro-batch (gradient
    accumulation over micro-batches on separate HIP streams, the reference's
    ``virtual_minibatch_size``, tssep/train/experiment.py:135-151): every micro-batch accumulates
    into its own buffer (no read-modify-write races between streams) and ``reduce_replicas`` folds
    them into buffer 0 = ``p.grad``."""

    def __init__(self, params, replicas=1):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev, dt = self.params[0].device, self.params[0].dtype
        self.flats = [torch.zeros(n, device=dev, dtype=dt) for _ in range(replicas)]
        self.flat = self.flats[0]
        off = 0
        for p in self.params:
            views = [f[off:off + p.numel()].view_as(p) for f in self.flats]
            p.grad = views[0]
            # the HIP backward may accumulate straight into these views (off the autograd engine,
            # on a side stream): see tssep_amd.functional._grad_sink
            p._tssep_grad_sinks = views
            p._tssep_grad_sink = views[0]
            off += p.numel()

    def reduce_replicas(self):
        for f in self.flats[1:]:
            self.flat.add_(f)

    def zero(self):
        for f in self.flats:
            f.zero_()

    def sync(self):
        """Wait for gradient work queued on the side stream (no-op on CPU / when unused)."""
        if self.flat.is_cuda:
            from . import hip_ops
            hip_ops.join_side_stream(self.flat.device)

    def all_reduce(self, group=None, async_op=False):
        self.sync()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return None

    def global_norm(self):
        return torch.linalg.vector_norm(self.flat)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items units for `rank` (pure data parallel)."""
    per, rem = divmod(n_items, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)
