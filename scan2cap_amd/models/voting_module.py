"""Hough voting head, drop-in for models/voting_module.py:11-60."""
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2 import fused


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

    def forward(self, seed_xyz, seed_features):
        """seed_xyz (B,S,3), seed_features (B,C,S) ->
        vote_xyz (B,S*vf,3), vote_features (B,C,S*vf)."""
        B, S = seed_xyz.shape[0], seed_xyz.shape[1]
        V = S * self.vote_factor
        if seed_features.is_cuda and self.in_dim % 4 == 0:
            # point-major rows through the fused GEMM/BN kernels
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
            net = fused.mlp_rows(rows, specs, params)
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
