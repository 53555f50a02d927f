"""Autograd wrappers and groupers over the HIP operator layer -- the drop-in for
lib/pointnet2/pointnet2_utils.py (FurthestPointSampling :51-80, GatherOperation
:83-117, ThreeNN :120-149, ThreeInterpolate :152-206, GroupingOperation
:209-257, BallQuery :260-291, QueryAndGroup :294-376, GroupAll :379-425).
Same call signatures (note ball_query(radius, nsample, xyz, new_xyz)), same
differentiability: FPS / ball_query / three_nn outputs carry no gradient.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext
from . import ops as _ops  # noqa: F401  (registers torch.ops.s2c.*)


def _op(name, ref):
    """CUDA tensors go through the registered custom op `torch.ops.s2c.<name>`; anything
    else reaches `_ext.<name>`, which rejects it like the reference ("CPU not supported",
    ball_query.cpp:27-29) -- or is the oracle when a CPU test has injected it there."""
    if ref.is_cuda:
        return getattr(torch.ops.s2c, name)
    return getattr(_ext, name)


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        inds = _op("furthest_point_sampling", xyz)(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n = features.size(2)
        ctx.save_for_backward(idx)
        return _op("gather_points", features)(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        g = grad_out.contiguous()
        return _op("gather_points_grad", g)(g, idx, ctx.n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _op("three_nn", unknown)(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx  # pointnet2_utils.py:142

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.m = features.size(2)
        ctx.save_for_backward(idx, weight)
        return _op("three_interpolate", features)(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        go = grad_out.contiguous()
        g = _op("three_interpolate_grad", go)(go, idx, weight, ctx.m)
        return g, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n = features.size(2)
        ctx.save_for_backward(idx)
        return _op("group_points", features)(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        g = grad_out.contiguous()
        return _op("group_points_grad", g)(g, idx, ctx.n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = _op("ball_query", new_xyz)(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


def uniform_resample(idx, nsample):
    """`sample_uniformly` of the reference (pointnet2_utils.py:336-345): every ball's index
    row is replaced by its sorted distinct ids followed by `nsample - n_unique` of them drawn
    with replacement.  The reference loops over all B*npoint rows on the host with
    `torch.unique`; here the distinct ids are found for all rows at once on the device
    (sort + first-occurrence compaction) and only the draws stay on the host: ONE
    `torch.randint(0, n_unique, (nsample - n_unique,))` per row from the global CPU
    generator, in the reference's row order, so that the same `torch.manual_seed` gives the
    same groups.  Returns (idx (B,npoint,nsample) int32, unique_cnt (B,npoint) float32 on the
    CPU, as the reference's `torch.zeros((B, npoint))`)."""
    B, P, S = idx.shape
    assert S == nsample
    srt, _ = torch.sort(idx.long(), dim=-1)
    first = torch.ones_like(srt, dtype=torch.bool)
    first[..., 1:] = srt[..., 1:] != srt[..., :-1]
    n_unique = first.sum(-1)                                           # (B,P)
    # compact the distinct ids to the front of each row (stable: they stay sorted)
    slot = torch.where(first, torch.cumsum(first.long(), -1) - 1, torch.full_like(srt, S))
    uniq = torch.zeros(B, P, S + 1, dtype=torch.long, device=idx.device)
    uniq.scatter_(-1, slot, srt)
    counts = n_unique.cpu()
    pick = torch.arange(S).repeat(B * P, 1)                            # host
    flat = counts.view(-1).tolist()
    for r, nu in enumerate(flat):                                      # the reference's order
        pick[r, nu:] = torch.randint(0, nu, (S - nu,), dtype=torch.long)
    pick = pick.view(B, P, S).to(idx.device)
    out = torch.gather(uniq[..., :S], -1, pick).to(idx.dtype)
    return out, counts.to(torch.float32)


class QueryAndGroup(nn.Module):
    """Ball query + grouping (pointnet2_utils.py:294-376).

    Returns (B, 3+C, npoint, nsample) [, grouped_xyz (B,3,npoint,nsample)]
    [, unique_cnt (B,npoint)].
    The centring / radius normalisation are two separate ops exactly as the
    reference (:350, :352): (p - c) / r is not bit-equal to (p - c) * (1/r).
    `sample_uniformly` (:336-345; never enabled by CapNet) goes through
    `uniform_resample`.
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False,
                 normalize_xyz=False, sample_uniformly=False,
                 ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = None
        if self.sample_uniformly:
            idx, unique_cnt = uniform_resample(idx, self.nsample)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B,3,npoint,nsample)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            grouped_features = grouping_operation(features.contiguous(), idx)
            new_features = (torch.cat([grouped_xyz, grouped_features], dim=1)
                            if self.use_xyz else grouped_features)
        else:
            assert self.use_xyz, \
                "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """pointnet2_utils.py:379-425."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            new_features = (torch.cat([grouped_xyz, grouped_features], dim=1)
                            if self.use_xyz else grouped_features)
        else:
            new_features = grouped_xyz
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
