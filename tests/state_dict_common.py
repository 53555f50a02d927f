"""CapNet constructor variants whose state_dict layout is pinned (scripts/train.py:56-123)."""
_BASE = dict(num_class=18, num_heading_bin=1, num_size_cluster=18, input_feature_dim=4,
             num_proposal=256)

VARIANTS = {
    # scripts/train.py defaults with --use_topdown --use_relation --num_graph_steps 2 --num_locals 10
    "topdown_relation": dict(_BASE, num_locals=10, use_topdown=True, query_mode="corner",
                             graph_mode="edge_conv", num_graph_steps=2, use_relation=True),
    "topdown_relation_orientation_distance": dict(
        _BASE, input_feature_dim=132, num_locals=10, use_topdown=True, graph_mode="edge_conv",
        num_graph_steps=2, use_relation=True, use_orientation=True, use_distance=True, num_bins=6),
    "topdown_only": dict(_BASE, num_locals=-1, use_topdown=True, num_graph_steps=0),
    "plain_captioner": dict(_BASE, use_topdown=False, num_graph_steps=0),
    "detector_only": dict(_BASE, no_caption=True, num_graph_steps=0),
}
