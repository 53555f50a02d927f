"""Seeded inputs of the module variants CapNet's default configuration does not switch on:
GraphModule(graph_mode="graph_conv"), EdgeConv aggregation "mean" / "max"
(models/graph_module.py:117-151) and QueryAndGroup(sample_uniformly, ret_unique_cnt)
(lib/pointnet2/pointnet2_utils.py:336-345).  Shared by tests/gen_golden_variants.py (runs the
REFERENCE's modules) and tests/test_variants.py (runs scan2cap_amd's)."""
import numpy as np
import torch

GRAPH_CASES = {
    # name: GraphModule kwargs
    "graph_conv": dict(graph_mode="graph_conv", graph_aggr="add", return_orientation=False),
    "edge_add": dict(graph_mode="edge_conv", graph_aggr="add", return_orientation=True),
    "edge_mean": dict(graph_mode="edge_conv", graph_aggr="mean", return_orientation=True),
    "edge_max": dict(graph_mode="edge_conv", graph_aggr="max", return_orientation=True),
}
GRAPH_DIMS = dict(in_size=32, out_size=32, num_layers=2, num_proposals=24, feat_size=32,
                  num_locals=5, query_mode="corner")
GRAPH_OUT_KEYS = ("bbox_feature", "adjacent_mat", "edge_index", "edge_feature",
                  "num_edge_source", "num_edge_target", "edge_orientations", "edge_distances")


TIE_FREE_SCENES = slice(0, 2)      # every target there has >= num_locals finite candidates


def graph_inputs(seed=3, B=3, K=24, F=32):
    g = np.random.Generator(np.random.PCG64(seed))
    centers = g.uniform(-2.0, 2.0, size=(B, K, 1, 3))
    half = g.uniform(0.1, 0.6, size=(B, K, 1, 3))
    signs = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)],
                     np.float64).reshape(1, 1, 8, 3)
    corners = centers + signs * half                                  # float64, as the pipeline
    mask = (g.uniform(size=(B, K)) < 0.7).astype(np.int64)
    mask[0] = 1                                                       # all valid
    mask[2, 3:] = 0                                                   # fewer objects than locals
    feats = g.standard_normal((B, K, F)).astype(np.float32)
    return {"bbox_corner": corners, "bbox_mask": mask, "bbox_feature": feats}


def fill_params(module, seed):
    """Deterministic weights by parameter order (both implementations register the same
    names in the same order)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(module.named_parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1))


UNIFORM = dict(radius=0.35, nsample=16, npoint=48, N=512, C=5, B=2, seed=11, rng=1234,
               mlp=[5, 16, 24])


def uniform_inputs():
    u = UNIFORM
    g = np.random.Generator(np.random.PCG64(u["seed"]))
    xyz = g.uniform(-1.0, 1.0, size=(u["B"], u["N"], 3)).astype(np.float32)
    xyz[:, :, 2] *= 0.3
    feats = g.standard_normal((u["B"], u["C"], u["N"])).astype(np.float32)
    return xyz, feats
