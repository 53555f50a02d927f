"""Set-abstraction and feature-propagation layers used by CapNet -- drop-in for
lib/pointnet2/pointnet2_modules.py: PointnetSAModuleVotes (:164-272) and
PointnetFPModule (:356-416).  (The MSG / LFP variants at :26-163, :274-496 are
not instantiated on the CapNet path.)
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ext
from . import fused
from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class PointnetSAModuleVotes(nn.Module):
    """FPS -> gather centres -> ball query + group -> SharedMLP -> pool."""

    def __init__(self, *, mlp: List[int], npoint: int = None,
                 radius: float = None, nsample: int = None, bn: bool = True,
                 use_xyz: bool = True, pooling: str = "max", sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False,
                 ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else (
            radius / 2 if radius is not None else None)
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3  # in place, like the reference (:206-207)
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

    def _fused_ok(self, xyz):
        if not (xyz.is_cuda and self.npoint is not None and self.pooling == "max"):
            return False
        if self.sample_uniformly:        # host-side draws (pointnet2_utils.uniform_resample)
            return False
        specs, params = fused.shared_mlp_specs(self.mlp_module)
        return len(specs) > 0 and specs[-1].bn is not None and specs[-1].relu \
            and fused.mlp_supported(specs, params)

    def geometry(self, xyz, inds=None):
        """The xyz-only part of the stage: FPS -> centres -> ball query.  It does
        not depend on features or weights, so a pipeline can compute it ahead of
        time on another stream (scan2cap_amd/pipeline.py)."""
        if inds is None:
            if getattr(self, "fps_input_in_pick_order", False) and xyz.is_cuda:
                # the input is the previous stage's centres: picks 0..npoint-1, proven on the
                # device instead of npoint-1 serial rounds (same result for any input)
                inds = _ext.furthest_point_sampling(xyz.contiguous(), self.npoint,
                                                    prefix_hint=True)
            else:
                inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
        idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
        return inds, new_xyz, idx

    def _forward_fused(self, xyz, features, inds, geom=None):
        """MI355X path: point-major rows, no (B,C,npoint,nsample) tensors, no
        transposes (the reference flips xyz twice around gather_points,
        pointnet2_modules.py:233-240).  `features` (B,C,N) is consumed through its
        (B,N,C) transposed VIEW, so a point-major producer costs no copy; the
        returned features are likewise a (B,C,npoint) view of point-major data."""
        if geom is not None:
            inds, new_xyz, idx = geom
        else:
            if inds is not None:
                assert inds.shape[1] == self.npoint
            inds, new_xyz, idx = self.geometry(xyz, inds)
        feats_pm = features.transpose(1, 2) if features is not None else None
        if not self.use_xyz:
            raise NotImplementedError("use_xyz=False is not on the CapNet path")
        out = fused.sa_group_mlp_pool(xyz, new_xyz, feats_pm, idx, self.radius,
                                      self.normalize_xyz, self.mlp_module)
        return new_xyz, out.transpose(1, 2), inds

    def forward(self, xyz, features=None, inds=None, geom=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3),
        new_features (B,mlp[-1],npoint), inds (B,npoint) int32.
        `geom` = precomputed (inds, new_xyz, idx) from `geometry()` (optional).
        With `ret_unique_cnt` a fourth output (B,npoint) follows."""
        if self._fused_ok(xyz):
            return self._forward_fused(xyz, features, inds, geom)
        if geom is not None:
            inds = geom[0]
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        else:
            assert inds.shape[1] == self.npoint
        new_xyz = pointnet2_utils.gather_operation(
            xyz_flipped, inds).transpose(1, 2).contiguous() \
            if self.npoint is not None else None

        unique_cnt = None
        if self.ret_unique_cnt:                       # pointnet2_modules.py:242-249
            grouped_features, grouped_xyz, unique_cnt = self.grouper(xyz, new_xyz, features)
        else:
            grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)
        new_features = self.mlp_module(grouped_features)  # (B,C',npoint,nsample)
        if self.pooling == "max":
            new_features = F.max_pool2d(
                new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == "avg":
            new_features = F.avg_pool2d(
                new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == "rbf":
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False)
                            / (self.sigma ** 2) / 2)
            new_features = torch.sum(new_features * rbf.unsqueeze(1), -1,
                                     keepdim=True) / float(self.nsample)
        new_features = new_features.squeeze(-1)
        if self.ret_unique_cnt:                       # pointnet2_modules.py:269-272
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


# feature propagation on point-major rows (csrc/s2c_sa.hip: fp_interp_rows)
FUSE_FP = True


class PointnetFPModule(nn.Module):
    """three_nn -> inverse-distance weights -> three_interpolate -> concat skip
    -> SharedMLP (pointnet2_modules.py:371-416)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    @staticmethod
    def geometry(unknown, known):
        """xyz-only part: 3-NN indices and inverse-distance weights
        (pointnet2_modules.py:394-397)."""
        dist, idx = pointnet2_utils.three_nn(unknown, known)
        dist_recip = 1.0 / (dist + 1e-8)
        norm = torch.sum(dist_recip, dim=2, keepdim=True)
        return idx, dist_recip / norm

    def forward(self, unknown, known, unknow_feats, known_feats, geom=None):
        if known is not None and known_feats.is_cuda and FUSE_FP:
            specs, params = fused.shared_mlp_specs(self.mlp)
            if fused.mlp_supported(specs, params):
                # point-major: interpolation + skip concat written straight into the
                # MLP's row operand (features arrive as transposed views of row data)
                idx, weight = geom if geom is not None else self.geometry(unknown, known)
                B, n = idx.shape[:2]
                rows = fused.fp_rows(
                    known_feats.transpose(1, 2),
                    unknow_feats.transpose(1, 2) if unknow_feats is not None else None,
                    idx, weight)
                out = fused.mlp_rows(rows, specs, params)
                return out.view(B, n, -1).transpose(1, 2)
        if known is not None:
            idx, weight = geom if geom is not None else self.geometry(unknown, known)
            interpolated = pointnet2_utils.three_interpolate(
                known_feats.contiguous(), idx, weight)
        else:
            interpolated = known_feats.expand(
                *known_feats.size()[0:2], unknown.size(1))
        new_features = (torch.cat([interpolated, unknow_feats], dim=1)
                        if unknow_feats is not None else interpolated)
        if new_features.is_cuda:
            specs, params = fused.shared_mlp_specs(self.mlp)
            if fused.mlp_supported(specs, params):
                B, C, n = new_features.shape
                rows = new_features.transpose(1, 2).reshape(B * n, C)
                out = fused.mlp_rows(rows, specs, params)
                return out.view(B, n, -1).transpose(1, 2)
        new_features = self.mlp(new_features.unsqueeze(-1))
        return new_features.squeeze(-1)
