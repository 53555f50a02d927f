"""Generates tests/golden/variants.npz by running the REFERENCE's modules (imported from
/root/reference through oracle/ref_harness.py, CPU) on the seeded inputs of
tests/variants_common.py: GraphModule in graph_conv mode and with EdgeConv aggregation
add / mean / max, QueryAndGroup + PointnetSAModuleVotes with sample_uniformly /
ret_unique_cnt.  The .npz holds expected outputs only (inputs and weights are seeded).

The graph variants pass through the harness's torch_geometric shim (PyG is un-vendored
and unpinned by the reference): parity unpinned at that boundary, see oracle/ref_harness.py.

    python tests/gen_golden_variants.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from tests import variants_common as vc  # noqa: E402


def main():
    ref = ref_harness.reference_modules()
    out = {}
    inp = vc.graph_inputs()
    for name, kw in vc.GRAPH_CASES.items():
        m = ref.graph.GraphModule(**vc.GRAPH_DIMS, **kw).eval()
        vc.fill_params(m, seed=17)
        dd = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
        with torch.no_grad():
            dd = m(dd)
        for k in vc.GRAPH_OUT_KEYS:
            out["graph/%s/%s" % (name, k)] = dd[k].detach().cpu().numpy()
        print(name, "sources", dd["num_edge_source"].tolist(), "targets",
              dd["num_edge_target"].tolist())
    u = vc.UNIFORM
    xyz, feats = (torch.from_numpy(a) for a in vc.uniform_inputs())
    new_xyz = xyz[:, :u["npoint"]].contiguous()
    grouper = ref.pointnet2_utils.QueryAndGroup(u["radius"], u["nsample"], use_xyz=True,
                                                ret_grouped_xyz=True, sample_uniformly=True,
                                                ret_unique_cnt=True)
    torch.manual_seed(u["rng"])
    new_features, grouped_xyz, unique_cnt = grouper(xyz, new_xyz, feats)
    out["uniform/new_features"] = new_features.numpy()
    out["uniform/grouped_xyz"] = grouped_xyz.numpy()
    out["uniform/unique_cnt"] = unique_cnt.numpy()
    print("unique counts: min %d max %d" % (unique_cnt.min(), unique_cnt.max()))
    sa = ref.pointnet2_modules.PointnetSAModuleVotes(
        mlp=list(u["mlp"]), npoint=u["npoint"], radius=u["radius"], nsample=u["nsample"],
        use_xyz=True, normalize_xyz=True, sample_uniformly=True, ret_unique_cnt=True).eval()
    vc.fill_params(sa, seed=23)
    torch.manual_seed(u["rng"])
    with torch.no_grad():
        sx, sf, si, sc = sa(xyz, feats)
    out["sa/new_xyz"], out["sa/new_features"] = sx.numpy(), sf.numpy()
    out["sa/inds"], out["sa/unique_cnt"] = si.numpy(), sc.numpy()
    path = os.path.join(HERE, "golden", "variants.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
