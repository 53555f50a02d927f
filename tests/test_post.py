"""parse_predictions: oracle and HIP path against the golden produced by
the reference's own lib/ap_helper.py (tests/gen_golden_post.py)."""
import os

import numpy as np
import pytest
import torch

from tests import post_common as pc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "parse_predictions.npz")


def _check(name, gold, res, pred_mask):
    np.testing.assert_array_equal(pred_mask, gold[name + "/pred_mask"])
    flat = pc.flatten(res)
    np.testing.assert_array_equal(flat["scene"], gold[name + "/scene"])
    np.testing.assert_array_equal(flat["cls"], gold[name + "/cls"])
    np.testing.assert_allclose(flat["corners"], gold[name + "/corners"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(flat["score"], gold[name + "/score"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", list(pc.POST_DICTS))
def test_oracle_matches_reference_golden(name):
    from oracle import post
    gold = np.load(GOLD)
    inputs = {k[3:]: gold[k] for k in gold.files if k.startswith("in/")}
    cfg = dict(pc.POST_DICTS[name], dataset_config=pc.dataset_config())
    res, mask = post.parse_predictions(inputs, cfg)
    assert mask.sum() > 0 and mask.sum() < mask.size
    _check(name, gold, res, mask)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(pc.POST_DICTS))
def test_hip_matches_reference_golden(name):
    from scan2cap_amd import ap_helper
    gold = np.load(GOLD)
    dev = torch.device("cuda")
    ep = {k[3:]: torch.from_numpy(gold[k]).to(dev) for k in gold.files if k.startswith("in/")}
    cfg = dict(pc.POST_DICTS[name], dataset_config=pc.dataset_config())
    res = ap_helper.parse_predictions(ep, cfg)
    _check(name, gold, res, ep["pred_mask"])


@pytest.mark.gpu
def test_hip_matches_oracle_larger_and_strided():
    """Bigger seeded case; the point cloud is read in place with its (3+C) row stride."""
    from oracle import post
    from scan2cap_amd import ap_helper
    old = (pc.B, pc.K, pc.N)
    pc.B, pc.K, pc.N = 3, 160, 30000
    try:
        inputs = pc.make_inputs(seed=23)
    finally:
        pc.B, pc.K, pc.N = old
    dev = torch.device("cuda")
    for name in ("predict", "nms2d"):
        cfg = dict(pc.POST_DICTS[name], dataset_config=pc.dataset_config())
        ref, mask = post.parse_predictions(inputs, cfg)
        ep = {k: torch.from_numpy(v).to(dev) for k, v in inputs.items()}
        res = ap_helper.parse_predictions(ep, cfg)
        np.testing.assert_array_equal(ep["pred_mask"], mask)
        a, b = pc.flatten(res), pc.flatten(ref)
        np.testing.assert_array_equal(a["cls"], b["cls"])
        np.testing.assert_allclose(a["corners"], b["corners"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(a["score"], b["score"], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_parse_predictions_rejects_cpu_tensors():
    from scan2cap_amd import ap_helper
    inputs = pc.make_inputs(seed=1)
    ep = {k: torch.from_numpy(v) for k, v in inputs.items()}
    cfg = dict(pc.POST_DICTS["predict"], dataset_config=pc.dataset_config())
    with pytest.raises(RuntimeError):
        ap_helper.parse_predictions(ep, cfg)
