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


def dist_backend(backend=None):
    """Collective backend of the N>1 path: RCCL ("nccl" on ROCm) when there is a GPU per
    rank; `S2C_DIST_BACKEND=gloo` runs the very same launcher / two-graph step with several
    ranks sharing ONE GPU (RCCL refuses two ranks on one device), which is how the N>1 path
    is exercised on a single leased MI355X (tests/test_bench_launch_gpu.py)."""
    if backend is None:
        backend = os.environ.get("S2C_DIST_BACKEND")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"  # nccl == RCCL on ROCm
    return backend


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (as set by
    `python -m torch.distributed.run` or bench.py's own launcher).
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ranks of THIS node (torch.distributed.run exports LOCAL_WORLD_SIZE; a multi-node job with
    # fewer than 8 GPUs per node is as valid as 8 ranks on one node)
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "0"))
    if not local_world:
        # srun / mpirun with a hand-set RANK / WORLD_SIZE (e.g. 16 ranks on 2 x 8 GPUs) exports no
        # LOCAL_WORLD_SIZE: assume at most one rank per visible device on this node, and call the
        # launch over-subscribed only when LOCAL_RANK does not fit the devices
        local_world = min(world, max(ndev, 1))
        if ndev and local_rank >= ndev:
            local_world = local_rank + 1
        multi_node = any(os.environ.get(k) for k in ("SLURM_NNODES", "GROUP_RANK", "NODE_RANK",
                                                     "OMPI_COMM_WORLD_LOCAL_SIZE"))
        if "LOCAL_RANK" not in os.environ and world > max(ndev, 1) and not multi_node:
            # a hand-launched single-node job that sets only RANK / WORLD_SIZE with more ranks than
            # devices: every rank would default to device 0 -> over-subscribed (the persistent decoder
            # is switched off below, RCCL gets the clear error instead of its duplicate-GPU one)
            local_world = world
        if world > 1 and "LOCAL_RANK" not in os.environ:
            import warnings
            warnings.warn("scan2cap_amd.parallel: WORLD_SIZE=%d without LOCAL_RANK / LOCAL_WORLD_SIZE; "
                          "assuming %d rank(s) on this node's %d device(s)" % (world, local_world, ndev))
    if world > 1 and ndev and local_world > ndev:
        # several ranks share a device (the gloo rehearsal): two persistent decoder kernels of
        # different processes could each hold part of the CUs and wait for the rest
        from .models import decoder_fused
        decoder_fused.set_persist(False)
    if world > 1 and not dist.is_initialized():
        backend = dist_backend(backend)
        if backend == "nccl" and ndev and ndev < local_world:
            raise RuntimeError("RCCL needs one GPU per rank (%d ranks on this node, %d visible "
                               "GPUs); set S2C_DIST_BACKEND=gloo to run several ranks on one GPU"
                               % (local_world, ndev))
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


def split_detector_captioner(model):
    """Parameter groups in the order their gradients complete in backward: the captioner +
    relation graph first (~80 % of CapNet's parameter bytes: GRUs, classifier), the detector
    (backbone, voting, proposal) last."""
    late = [p for n in ("backbone_net", "vgen", "proposal") if hasattr(model, n)
            for p in getattr(model, n).parameters() if p.requires_grad]
    late_ids = {id(p) for p in late}
    early = [p for p in model.parameters() if p.requires_grad and id(p) not in late_ids]
    return early, late


class BucketedGradAllReduce(object):
    """Several flat buckets, reduced independently: bucket i can be on the wire (RCCL's own
    stream, `async_op=True`) while the backward pass of the later buckets still runs.
    Same drop / pack / reduce protocol as FlatGradAllReduce, per bucket."""

    def __init__(self, module, groups, process_group=None, broadcast=True):
        self.module, self.group = module, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.groups = [list(g) for g in groups]
        self.params = [p for g in self.groups for p in g]
        assert len({id(p) for p in self.params}) == len(self.params)
        assert {id(p) for p in self.params} == {id(p) for p in module.parameters()
                                                if p.requires_grad}, "groups must partition"
        dev = self.params[0].device
        self.flats = [torch.zeros(sum(p.numel() for p in g), dtype=torch.float32, device=dev)
                      for g in self.groups]
        self._work = [None] * len(self.groups)
        if broadcast and self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=0, group=self.group)

    def drop_grads(self):
        for p in self.params:
            p.grad = None

    def pack_grads(self, i):
        views, grads, off = [], [], 0
        flat = self.flats[i]
        for p in self.groups[i]:
            n = p.numel()
            v = flat[off:off + n].view_as(p)
            off += n
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if views:
            torch._foreach_copy_(views, grads)

    def reduce(self, i, async_op=True):
        """Average bucket i across the ranks.  async_op: the collective is enqueued behind
        the work already on the current stream and runs on the process group's stream;
        `wait(i)` orders the current stream behind it."""
        if self.world <= 1:
            return
        self.flats[i].div_(self.world)
        self._work[i] = dist.all_reduce(self.flats[i], op=dist.ReduceOp.SUM, group=self.group,
                                        async_op=async_op)

    def wait(self, i=None):
        for k in (range(len(self.groups)) if i is None else (i,)):
            if self._work[k] is not None:
                self._work[k].wait()
                self._work[k] = None

    @property
    def nbytes(self):
        return sum(f.numel() for f in self.flats) * 4


class TwoStageBackward(object):
    """backward_in_two_stages as two separately callable halves (each half is captured in
    its own hipGraph by bench.py, with the early bucket's all-reduce launched in between)."""

    def __init__(self, early_params, late_params):
        self.early, self.late = list(early_params), list(late_params)
        self._pending = None

    def stage1(self, data_dict):
        det, rest = data_dict.get("_loss_det"), data_dict.get("_loss_rest")
        X = data_dict.get("aggregated_vote_features")
        if det is None or rest is None or X is None or not rest.requires_grad \
                or not self.early or not X.requires_grad:
            data_dict["loss"].backward()
            self._pending = None
            return
        torch.autograd.backward([rest], inputs=self.early + [X], retain_graph=True)
        dX = X.grad
        X.grad = None
        self._pending = (det, X, dX)

    def stage2(self):
        if self._pending is None:
            return
        det, X, dX = self._pending
        self._pending = None
        roots, seeds = [det], [None]
        if dX is not None:
            roots.append(X)
            seeds.append(dX)
        torch.autograd.backward(roots, seeds, inputs=self.late)


def backward_in_two_stages(data_dict, early_params, late_params, between=None):
    """loss = det + rest, where `rest` (caption + relation losses) reaches the detector only
    through X = data_dict["aggregated_vote_features"] (the proposal features the graph and
    caption modules consume; box corners / masks / target ids are non-differentiable).
    Stage 1 differentiates `rest` down to X and the captioner / graph parameters;
    `between()` runs (e.g. starts their all-reduce); stage 2 carries d rest / d X and `det`
    through the detector.  The parameter gradients equal those of loss.backward()."""
    det, rest = data_dict.get("_loss_det"), data_dict.get("_loss_rest")
    X = data_dict.get("aggregated_vote_features")
    if det is None or rest is None or X is None or not rest.requires_grad \
            or not early_params or not X.requires_grad:
        data_dict["loss"].backward()
        if between is not None:
            between()
        return
    torch.autograd.backward([rest], inputs=list(early_params) + [X], retain_graph=True)
    dX = X.grad
    X.grad = None
    if between is not None:
        between()
    roots, seeds = [det], [None]
    if dX is not None:
        roots.append(X)
        seeds.append(dX)
    torch.autograd.backward(roots, seeds, inputs=list(late_params))
