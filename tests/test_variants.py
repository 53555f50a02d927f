"""Module variants outside CapNet's default configuration -- GraphModule in graph_conv mode,
EdgeConv aggregation mean / max, QueryAndGroup(sample_uniformly, ret_unique_cnt) and the
4-output PointnetSAModuleVotes -- against tests/golden/variants.npz, produced by the
reference's own classes (tests/gen_golden_variants.py).  CPU (oracle ops as the test double
of the extension) and GPU (HIP)."""
import os

import numpy as np
import pytest
import torch

from tests import variants_common as vc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "variants.npz")


@pytest.fixture
def cpu_ops():
    """The CPU oracle injected as scan2cap_amd.pointnet2._ext (host-logic tests)."""
    from oracle import torch_ext
    from scan2cap_amd.pointnet2 import _ext
    saved = {n: getattr(_ext, n) for n in torch_ext.NAMES}
    for n in torch_ext.NAMES:
        setattr(_ext, n, getattr(torch_ext, n))
    yield
    for n, f in saved.items():
        setattr(_ext, n, f)


def _graph(device, name):
    from scan2cap_amd.models.graph_module import GraphModule
    gold = np.load(GOLD)
    m = GraphModule(**vc.GRAPH_DIMS, **vc.GRAPH_CASES[name]).eval()
    vc.fill_params(m, seed=17)
    m = m.to(device)
    dd = {k: torch.from_numpy(v.copy()).to(device) for k, v in vc.graph_inputs().items()}
    with torch.no_grad():
        dd = m(dd)
    # scene 2 has 3 valid objects for num_locals = 5: the 5 "nearest" then include +inf
    # entries and WHICH of them is torch.topk's unspecified tie order (the reference's CPU
    # and CUDA builds differ there too; csrc/s2c_graph.hip takes the smallest ids).  The
    # CPU path shares torch.topk with the golden's producer and is compared on every scene,
    # the device path on the tie-free ones.
    scenes = slice(None) if device.type == "cpu" else vc.TIE_FREE_SCENES
    for k in vc.GRAPH_OUT_KEYS:
        want = gold["graph/%s/%s" % (name, k)][scenes]
        got = dd[k].detach().cpu().numpy()[scenes]
        assert got.shape == want.shape, k
        if want.dtype.kind in "iu" or k in ("adjacent_mat", "edge_index"):
            np.testing.assert_array_equal(got, want, err_msg=k)
        else:
            scale = max(1.0, float(np.abs(want).max()))
            assert float(np.abs(got - want).max()) <= 1e-4 * scale, k
    if device.type != "cpu":
        # the tie scene: same edge counts, finite outputs, rows of invalid objects untouched
        np.testing.assert_array_equal(dd["num_edge_source"].cpu().numpy(),
                                      gold["graph/%s/num_edge_source" % name])
        assert torch.isfinite(dd["bbox_feature"]).all()
        inv = vc.graph_inputs()["bbox_mask"][2] == 0
        assert float(dd["bbox_feature"][2][torch.from_numpy(inv).to(device)].abs().max()) == 0.0


@pytest.mark.parametrize("name", sorted(vc.GRAPH_CASES))
def test_graph_variants_cpu(name):
    _graph(torch.device("cpu"), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(vc.GRAPH_CASES))
def test_graph_variants_gpu(name):
    _graph(torch.device("cuda"), name)


@pytest.mark.gpu
def test_edge_aggregations_differentiate():
    """mean / max re-use the HIP scatter kernel's sum and masked messages: gradients
    against the plain torch restatement of the same layer."""
    from scan2cap_amd.models import graph_module as gm
    dev = torch.device("cuda")
    inp = vc.graph_inputs()
    for aggr in ("add", "mean", "max"):
        grads = []
        for kernels in (True, False):
            m = gm.GraphModule(**vc.GRAPH_DIMS, graph_mode="edge_conv", graph_aggr=aggr,
                               return_orientation=True)
            vc.fill_params(m, seed=5)
            m = m.to(dev)
            dd = {k: torch.from_numpy(v.copy()).to(dev) for k, v in inp.items()}
            dd["bbox_feature"].requires_grad_(True)
            x = dd["bbox_feature"]
            saved = gm.USE_EDGE_KERNELS
            gm.USE_EDGE_KERNELS = kernels
            try:
                out = m(dd)
                (out["bbox_feature"].square().sum() + out["edge_orientations"].sum()).backward()
            finally:
                gm.USE_EDGE_KERNELS = saved
            grads.append([x.grad.clone()] + [p.grad.clone() for _, p in sorted(m.named_parameters())])
        for a, b in zip(*grads):
            scale = max(1.0, float(b.abs().max()))
            assert float((a - b).abs().max()) <= 1e-4 * scale, aggr


def _uniform(device):
    from scan2cap_amd.pointnet2 import pointnet2_utils
    from scan2cap_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    gold = np.load(GOLD)
    u = vc.UNIFORM
    xyz, feats = (torch.from_numpy(a).to(device) for a in vc.uniform_inputs())
    new_xyz = xyz[:, :u["npoint"]].contiguous()
    grouper = pointnet2_utils.QueryAndGroup(u["radius"], u["nsample"], use_xyz=True,
                                            ret_grouped_xyz=True, sample_uniformly=True,
                                            ret_unique_cnt=True)
    torch.manual_seed(u["rng"])          # the draws come from the global CPU generator
    new_features, grouped_xyz, unique_cnt = grouper(xyz, new_xyz, feats)
    assert unique_cnt.device.type == "cpu"          # pointnet2_utils.py:337: torch.zeros(...)
    np.testing.assert_array_equal(unique_cnt.numpy(), gold["uniform/unique_cnt"])
    # same ids drawn -> the gathered values are the same floats
    np.testing.assert_array_equal(grouped_xyz.cpu().numpy(), gold["uniform/grouped_xyz"])
    np.testing.assert_array_equal(new_features.cpu().numpy(), gold["uniform/new_features"])
    sa = PointnetSAModuleVotes(mlp=list(u["mlp"]), npoint=u["npoint"], radius=u["radius"],
                               nsample=u["nsample"], use_xyz=True, normalize_xyz=True,
                               sample_uniformly=True, ret_unique_cnt=True).eval()
    vc.fill_params(sa, seed=23)
    sa = sa.to(device)
    torch.manual_seed(u["rng"])
    with torch.no_grad():
        sx, sf, si, sc = sa(xyz, feats)
    np.testing.assert_array_equal(si.cpu().numpy(), gold["sa/inds"])
    np.testing.assert_array_equal(sx.cpu().numpy(), gold["sa/new_xyz"])
    np.testing.assert_array_equal(sc.numpy(), gold["sa/unique_cnt"])
    np.testing.assert_allclose(sf.cpu().numpy(), gold["sa/new_features"], rtol=1e-4, atol=1e-5)


def test_sample_uniformly_cpu(cpu_ops):
    _uniform(torch.device("cpu"))


@pytest.mark.gpu
def test_sample_uniformly_gpu():
    _uniform(torch.device("cuda"))


def test_ret_unique_cnt_needs_sample_uniformly():
    from scan2cap_amd.pointnet2 import pointnet2_utils
    with pytest.raises(AssertionError):                   # pointnet2_utils.py:314-315
        pointnet2_utils.QueryAndGroup(0.2, 8, ret_unique_cnt=True)


def test_invalid_aggregation_is_rejected():
    from scan2cap_amd.models.graph_module import EdgeConv
    with pytest.raises(ValueError):
        EdgeConv(8, 8, "median")
