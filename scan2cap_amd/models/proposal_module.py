"""Vote aggregation + proposal head + box decode, drop-in for
models/proposal_module.py:21-144.

MI355X-first change: `decode_pred_box` stays on the device.  The reference
copies five tensors to the host, decodes in numpy float64 and copies the
corners back (:82-99) -- a full pipeline sync.  Here the same float64 arithmetic
runs as torch ops on the GPU (scan2cap_amd/box_util.py) and `bbox_corner` is
float64 as in the reference.
"""
import numpy as np
import torch
import torch.nn as nn

import ctypes

from .. import _C
from ..box_util import get_3d_box_batch
from ..pointnet2 import fused
from ..pointnet2.pointnet2_modules import PointnetSAModuleVotes

_I, _P = ctypes.c_int, ctypes.c_void_p
_C.register("s2c_proposal_decode", [_I] * 6 + [_P] * 9)
# box size / corners / arg-max bookkeeping in one launch (csrc/s2c_boxes.hip) instead of ~30
# framework micro-kernels; identical values (tests/test_fused_gpu.py)
FUSE_BOX_DECODE = True


class ProposalModule(nn.Module):
    def __init__(self, num_class, num_heading_bin, num_size_cluster,
                 mean_size_arr, num_proposal, sampling, seed_feat_dim=256):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_proposal = num_proposal
        self.sampling = sampling
        self.seed_feat_dim = seed_feat_dim
        self.vote_aggregation = PointnetSAModuleVotes(
            npoint=self.num_proposal, radius=0.3, nsample=16,
            mlp=[self.seed_feat_dim, 128, 128, 128],
            use_xyz=True, normalize_xyz=True)
        # objectness(2) + centre(3) + heading cls/res + size cls/res(4x) + sem cls
        nout = 2 + 3 + num_heading_bin * 2 + num_size_cluster * 4 + num_class
        self.proposal = nn.Sequential(
            nn.Conv1d(128, 128, 1, bias=False), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Conv1d(128, 128, 1, bias=False), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Conv1d(128, nout, 1))
        # device copies of the constants (non-persistent: not in the state_dict)
        msa = np.asarray(mean_size_arr)
        self.register_buffer("_mean_size_f32",
                             torch.from_numpy(msa.astype(np.float32)),
                             persistent=False)
        self.register_buffer("_mean_size_f64",
                             torch.from_numpy(msa.astype(np.float64)),
                             persistent=False)

    def forward(self, xyz, features, data_dict):
        xyz, features, fps_inds = self.vote_aggregation(xyz, features)
        data_dict["aggregated_vote_xyz"] = xyz
        data_dict["aggregated_vote_features"] = features.permute(0, 2, 1).contiguous()
        data_dict["aggregated_vote_inds"] = fps_inds
        if fused.fused_available(features):
            B, C, K = features.shape
            p = self.proposal
            specs = [fused.LayerSpec(False, p[1], True),
                     fused.LayerSpec(False, p[4], True),
                     fused.LayerSpec(True, None, False)]
            params = [p[0].weight.view(128, -1), p[1].weight, p[1].bias,
                      p[3].weight.view(128, -1), p[4].weight, p[4].bias,
                      p[6].weight.view(p[6].out_channels, -1), p[6].bias]
            rows = features.transpose(1, 2).reshape(B * K, C)
            net = fused.mlp_rows(rows, specs, params).view(B, K, -1).transpose(1, 2)
        else:
            net = self.proposal(features)
        return self.decode_scores(net, data_dict, self.num_class,
                                  self.num_heading_bin, self.num_size_cluster,
                                  self.mean_size_arr)

    def decode_pred_box(self, data_dict):
        """(B,K,8,3) float64 corners; arithmetic of proposal_module.py:80-103 +
        model_util_scannet.py:165-172 + box_util.py:360-383, on the device."""
        center = data_dict["center"].detach().double()
        size_class = torch.argmax(data_dict["size_scores"], -1)  # (B,K)
        res = torch.gather(
            data_dict["size_residuals"].detach(), 2,
            size_class.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, 1, 3)).squeeze(2)
        # class2size_batch: float64 mean size + float32 residual -> float64
        box_size = self._mean_size_f64[size_class] + res.double()
        # class2angle_batch is identically 0 for ScanNet; obb[:,6] = heading * -1
        heading = torch.zeros(center.shape[:2], dtype=torch.float64,
                              device=center.device) * -1
        return get_3d_box_batch(box_size, heading, center)

    def decode_scores(self, net, data_dict, num_class, num_heading_bin,
                      num_size_cluster, mean_size_arr):
        nt = net.transpose(2, 1).contiguous()  # (B,K,nout)
        B, K = nt.shape[0], nt.shape[1]
        NH, NS = num_heading_bin, num_size_cluster
        objectness_scores = nt[:, :, 0:2]
        center = data_dict["aggregated_vote_xyz"] + nt[:, :, 2:5]
        o = 5
        heading_scores = nt[:, :, o:o + NH]
        heading_residuals_normalized = nt[:, :, o + NH:o + NH * 2]
        o += NH * 2
        size_scores = nt[:, :, o:o + NS]
        size_residuals_normalized = nt[:, :, o + NS:o + NS * 4].view(B, K, NS, 3)
        sem_cls_scores = nt[:, :, o + NS * 4:]

        data_dict["_head_rows"] = nt       # the loss reads / differentiates the rows in place
        data_dict["objectness_scores"] = objectness_scores
        data_dict["center"] = center
        data_dict["heading_scores"] = heading_scores
        data_dict["heading_residuals_normalized"] = heading_residuals_normalized
        data_dict["heading_residuals"] = heading_residuals_normalized * (np.pi / NH)
        data_dict["size_scores"] = size_scores
        data_dict["size_residuals_normalized"] = size_residuals_normalized
        data_dict["size_residuals"] = size_residuals_normalized * \
            self._mean_size_f32.unsqueeze(0).unsqueeze(0)
        data_dict["sem_cls_scores"] = sem_cls_scores

        data_dict["bbox_feature"] = data_dict["aggregated_vote_features"]
        if FUSE_BOX_DECODE and nt.is_cuda and nt.dtype == torch.float32:
            # one launch for the non-differentiable bookkeeping (csrc/s2c_boxes.hip)
            dev = nt.device
            corners = torch.empty((B, K, 8, 3), dtype=torch.float64, device=dev)
            mask = torch.empty((B, K), dtype=torch.int64, device=dev)
            sem = torch.empty((B, K), dtype=torch.int64, device=dev)
            cen = center.detach()
            if not cen.is_contiguous():
                cen = cen.contiguous()
            with torch.cuda.device(dev):
                _C.call("s2c_proposal_decode", B, K, nt.shape[2], NH, NS, num_class,
                        nt.data_ptr(), cen.data_ptr(), self._mean_size_f32.data_ptr(),
                        self._mean_size_f64.data_ptr(), corners.data_ptr(), mask.data_ptr(),
                        sem.data_ptr(), None, _C.stream_ptr())
            data_dict["bbox_corner"] = corners
            data_dict["bbox_mask"] = mask
            data_dict["bbox_sems"] = sem
            data_dict["sem_cls"] = sem
            return data_dict
        data_dict["bbox_corner"] = self.decode_pred_box(data_dict)
        data_dict["bbox_mask"] = objectness_scores.argmax(-1)
        data_dict["bbox_sems"] = sem_cls_scores.argmax(-1)
        data_dict["sem_cls"] = sem_cls_scores.argmax(-1)
        return data_dict
