"""Hough voting head, drop-in for models/voting_module.py:11-60."""
import torch.nn as nn
import torch.nn.functional as F


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
        net = F.relu(self.bn1(self.conv1(seed_features)))
        net = F.relu(self.bn2(self.conv2(net)))
        net = self.conv3(net)  # (B,(3+C)*vf,S)
        net = net.transpose(2, 1).view(B, S, self.vote_factor, 3 + self.out_dim)
        vote_xyz = (seed_xyz.unsqueeze(2) + net[..., 0:3]).contiguous().view(B, V, 3)
        vote_features = seed_features.transpose(2, 1).unsqueeze(2) + net[..., 3:]
        vote_features = vote_features.contiguous().view(B, V, self.out_dim)
        return vote_xyz, vote_features.transpose(2, 1).contiguous()
