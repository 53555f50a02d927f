"""CapNet: detection (PointNet++/VoteNet) -> relational graph -> caption decoder.
Drop-in for models/capnet.py:15-123: same constructor signature, the same
re-assignable submodules (`backbone_net`, `vgen`, `proposal`, `graph`,
`caption`; scripts/train.py:103-105 mounts pretrained ones), the same
`forward(data_dict, use_tf=True, is_eval=False) -> data_dict` contract and
state_dict names.
"""
import torch
import torch.nn as nn

from .backbone_module import Pointnet2Backbone
from .caption_module import SceneCaptionModule, TopDownSceneCaptionModule
from .graph_module import GraphModule
from .proposal_module import ProposalModule
from .voting_module import VotingModule


class CapNet(nn.Module):
    def __init__(self, num_class, vocabulary, embeddings, num_heading_bin,
                 num_size_cluster, mean_size_arr, input_feature_dim=0,
                 num_proposal=256, num_locals=-1, vote_factor=1,
                 sampling="vote_fps", no_caption=False, use_topdown=False,
                 query_mode="corner", graph_mode="graph_conv",
                 num_graph_steps=0, use_relation=False, graph_aggr="add",
                 use_orientation=False, num_bins=6, use_distance=False,
                 use_new=False, emb_size=300, hidden_size=512):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        assert mean_size_arr.shape[0] == self.num_size_cluster
        self.input_feature_dim = input_feature_dim
        self.num_proposal = num_proposal
        self.vote_factor = vote_factor
        self.sampling = sampling
        self.no_caption = no_caption
        self.num_graph_steps = num_graph_steps

        self.backbone_net = Pointnet2Backbone(input_feature_dim=self.input_feature_dim)
        self.vgen = VotingModule(self.vote_factor, 256)
        self.proposal = ProposalModule(num_class, num_heading_bin, num_size_cluster,
                                       mean_size_arr, num_proposal, sampling)
        if use_relation:
            assert use_topdown  # relations only feed the top-down captioner
        if num_graph_steps > 0:
            self.graph = GraphModule(
                128, 128, num_graph_steps, num_proposal, 128, num_locals,
                query_mode, graph_mode, return_edge=use_relation,
                graph_aggr=graph_aggr, return_orientation=use_orientation,
                num_bins=num_bins, return_distance=use_distance)
        if not no_caption:
            if use_topdown:
                self.caption = TopDownSceneCaptionModule(
                    vocabulary, embeddings, emb_size, 128, hidden_size,
                    num_proposal, num_locals, query_mode, use_relation)
            else:
                self.caption = SceneCaptionModule(
                    vocabulary, embeddings, emb_size, 128, hidden_size, num_proposal)

    # Stage outputs are published under the reference's key names (SURVEY Appendix A:
    # backbone -> seeds -> votes -> proposals -> graph -> captions); the stages
    # themselves are the re-assignable sub-modules.
    _SEED_KEYS = (("seed_inds", "fp2_inds"), ("seed_xyz", "fp2_xyz"),
                  ("seed_features", "fp2_features"))

    def _host_prologue(self, data_dict, is_eval):
        """The teacher-forced decoder needs max(lang_len) as a host integer: read it
        before anything is enqueued (callers that know it pass `_num_words` and avoid
        the device read altogether)."""
        wants = not (self.no_caption or is_eval)
        if wants and "_num_words" not in data_dict and "lang_len" in data_dict:
            data_dict["_num_words"] = int(data_dict["lang_len"].max())

    def detect(self, data_dict):
        """capnet.py:86-109: backbone, vote generation (L2-normalised vote features),
        proposal aggregation + box heads."""
        data_dict = self.backbone_net(data_dict)
        for dst, src in self._SEED_KEYS:
            data_dict[dst] = data_dict[src]
        if hasattr(self.vgen, "forward_normalized"):      # offsets + L2 norm in one kernel
            vote_xyz, vote_feat = self.vgen.forward_normalized(data_dict["seed_xyz"],
                                                               data_dict["seed_features"])
        else:
            vote_xyz, vote_feat = self.vgen(data_dict["seed_xyz"], data_dict["seed_features"])
            vote_feat = vote_feat.div(torch.norm(vote_feat, p=2, dim=1).unsqueeze(1))
        data_dict.update(vote_xyz=vote_xyz, vote_features=vote_feat)
        return self.proposal(vote_xyz, vote_feat, data_dict)

    def forward(self, data_dict, use_tf=True, is_eval=False):
        self._host_prologue(data_dict, is_eval)
        data_dict = self.detect(data_dict)
        if self.num_graph_steps > 0:                     # relational graph (:111-116)
            data_dict = self.graph(data_dict)
        if not self.no_caption:                          # captioner (:118-121)
            data_dict = self.caption(data_dict, use_tf, is_eval)
        return data_dict
