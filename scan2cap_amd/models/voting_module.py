"""Hough voting head, drop-in for models/voting_module.py:11-60."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from .. import _C
from ..pointnet2 import fused


_I, _L, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p
_C.register("s2c_vote_head_fwd", [_I, _I, _P, _P, _P, _L, _P, _P, _P, _P])
_C.register("s2c_vote_head_bwd", [_I, _I, _P, _P, _L, _L, _P, _P, _P, _P, _P])
FUSE_VOTE_HEAD = True


class _VoteHead(Function):
    """(net (M,3+C), seed_xyz (M,3), seed rows (M,C)) -> vote_xyz (M,3), normalised vote
    features (M,C): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, net, seed_xyz, seed_rows):
        M, C = seed_rows.shape
        dev = net.device
        net = net if net.is_contiguous() else net.contiguous()
        seed_xyz = seed_xyz if seed_xyz.is_contiguous() else seed_xyz.contiguous()
        if seed_rows.stride(1) != 1:
            seed_rows = seed_rows.contiguous()
        vote_xyz = torch.empty((M, 3), dtype=torch.float32, device=dev)
        y = torch.empty((M, C), dtype=torch.float32, device=dev)
        norm = torch.empty(M, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _C.call("s2c_vote_head_fwd", M, C, net.data_ptr(), seed_xyz.data_ptr(),
                    seed_rows.data_ptr(), seed_rows.stride(0), vote_xyz.data_ptr(),
                    y.data_ptr(), norm.data_ptr(), _C.stream_ptr())
        ctx.save_for_backward(y, norm)
        return vote_xyz, y

    @staticmethod
    def backward(ctx, g_xyz, g_y):
        y, norm = ctx.saved_tensors
        M, C = y.shape
        dev = y.device
        if g_y is None:
            g_y = torch.zeros_like(y)
        if g_xyz is not None and not g_xyz.is_contiguous():
            g_xyz = g_xyz.contiguous()
        d_net = torch.empty((M, 3 + C), dtype=torch.float32, device=dev)
        d_seed = torch.empty((M, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _C.call("s2c_vote_head_bwd", M, C, g_xyz.data_ptr() if g_xyz is not None else None,
                    g_y.data_ptr(), g_y.stride(0), g_y.stride(1), y.data_ptr(),
                    norm.data_ptr(), d_net.data_ptr(), d_seed.data_ptr(), _C.stream_ptr())
        d_seed_xyz = g_xyz if ctx.needs_input_grad[1] else None
        return d_net, d_seed_xyz, d_seed


class VotingModule(nn.Module):
    def __init__(self, vote_factor, seed_feature_dim):
        super().__init__()
        self.vote_factor = vote_factor
        self.in_dim = seed_feature_dim
        self.out_dim = self.in_dim  # residual features: in == out
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv3 = nn.Conv1d(self.in_dim, (3 + self.out_dim) * self.vote_factor, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.bn2 = nn.BatchNorm1d(self.in_dim)

    def _net(self, seed_features, B, S):
        """Vote MLP on point-major rows -> (B*S, vote_factor*(3+C)), and the seed rows."""
        specs = [fused.LayerSpec(True, self.bn1, True),
                 fused.LayerSpec(True, self.bn2, True),
                 fused.LayerSpec(True, None, False)]
        params = [self.conv1.weight.view(self.in_dim, -1), self.conv1.bias,
                  self.bn1.weight, self.bn1.bias,
                  self.conv2.weight.view(self.in_dim, -1), self.conv2.bias,
                  self.bn2.weight, self.bn2.bias,
                  self.conv3.weight.view(self.conv3.out_channels, -1),
                  self.conv3.bias]
        rows = seed_features.transpose(2, 1).reshape(B * S, self.in_dim)
        return fused.mlp_rows(rows, specs, params), rows

    def forward_normalized(self, seed_xyz, seed_features):
        """forward() followed by the L2 normalisation of the vote features that CapNet applies
        (models/capnet.py:97-98), with offsets / residual / normalisation in one kernel
        (csrc/s2c_boxes.hip) when the fused path applies."""
        B, S = seed_xyz.shape[0], seed_xyz.shape[1]
        if not (FUSE_VOTE_HEAD and fused.fused_available(seed_features) and self.in_dim % 4 == 0
                and self.vote_factor == 1 and self.out_dim == self.in_dim
                and seed_features.dtype == torch.float32):
            vote_xyz, f = self.forward(seed_xyz, seed_features)
            return vote_xyz, f.div(torch.norm(f, p=2, dim=1).unsqueeze(1))
        net, rows = self._net(seed_features, B, S)
        vote_xyz, y = _VoteHead.apply(net, seed_xyz.reshape(B * S, 3), rows)
        return vote_xyz.view(B, S, 3), y.view(B, S, self.out_dim).transpose(2, 1)

    def forward(self, seed_xyz, seed_features):
        """seed_xyz (B,S,3), seed_features (B,C,S) ->
        vote_xyz (B,S*vf,3), vote_features (B,C,S*vf)."""
        B, S = seed_xyz.shape[0], seed_xyz.shape[1]
        V = S * self.vote_factor
        if fused.fused_available(seed_features) and self.in_dim % 4 == 0:
            # point-major rows through the fused GEMM/BN kernels
            net, _ = self._net(seed_features, B, S)
            net = net.view(B, S, self.vote_factor, 3 + self.out_dim)
        else:
            net = F.relu(self.bn1(self.conv1(seed_features)))
            net = F.relu(self.bn2(self.conv2(net)))
            net = self.conv3(net)  # (B,(3+C)*vf,S)
            net = net.transpose(2, 1).view(B, S, self.vote_factor, 3 + self.out_dim)
        vote_xyz = (seed_xyz.unsqueeze(2) + net[..., 0:3]).contiguous().view(B, V, 3)
        vote_features = seed_features.transpose(2, 1).unsqueeze(2) + net[..., 3:]
        vote_features = vote_features.contiguous().view(B, V, self.out_dim)
        # (B,C,V) view of point-major data; the reference returns a contiguous copy
        return vote_xyz, vote_features.transpose(2, 1)
