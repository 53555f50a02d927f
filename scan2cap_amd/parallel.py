"""Data-parallel training over the GPUs of one node: one process per GPU,
scenes sharded across ranks, ONE gradient all-reduce per step over RCCL/xGMI.

The reference has no multi-GPU path at all (single `cuda:0`,
scripts/train.py:132; zero collective call sites).  Every stage of the hot path
is per-scene, so the only exchange step is the gradient average:
~6 M fp32 parameters = ~25 MB, which fits one flat bucket.  All parameter
gradients are *views into one contiguous buffer*, so the collective runs
in place on a single large message (xGMI is point-to-point: few large messages
beat many small ones) with no flatten/unflatten copies.

BatchNorm statistics stay per replica (as torch DDP's default); buffers are
broadcast from rank 0 once at wrap time.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (as set by
    `python -m torch.distributed.run`).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # nccl == RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of `total` scenes owned by `rank`."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradAllReduce(object):
    """Owns one flat fp32 gradient buffer; p.grad of every trainable parameter is
    a view into it.  `reduce()` averages it across ranks with one all-reduce."""

    def __init__(self, module, process_group=None, broadcast=True):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in module.parameters() if p.requires_grad]
        self.params = params
        total = sum(p.numel() for p in params)
        dev = params[0].device if params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        if broadcast and self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=0, group=self.group)

    def zero_grad(self):
        """Keep the views alive: zero the bucket instead of dropping .grad
        (autograd then ACCUMULATES into the views: one add_ per parameter)."""
        self.flat.zero_()

    def drop_grads(self):
        """Alternative to zero_grad(): let autograd produce fresh gradient tensors
        (no zero-fill, no accumulate kernels -- ~2 launches per parameter saved)
        and move them into the bucket afterwards with pack_grads()."""
        for p in self.params:
            p.grad = None

    def pack_grads(self):
        """Copy the freshly produced gradients into the flat bucket with one
        multi-tensor copy and re-bind p.grad to the bucket views."""
        views, grads = [], []
        off = 0
        for p in self.params:
            n = p.numel()
            v = self.flat[off:off + n].view_as(p)
            off += n
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if views:
            torch._foreach_copy_(views, grads)

    def reattach(self):
        """(Re)bind p.grad to the bucket, e.g. after optimizer.zero_grad(set_to_none=True)."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat[off:off + n].data_ptr():
                p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def reduce(self, async_op=False):
        if self.world <= 1:
            return None
        self.flat.div_(self.world)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group,
                               async_op=async_op)

    @property
    def nbytes(self):
        return self.flat.numel() * 4
