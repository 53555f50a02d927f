"""The teacher-forced decoder's inputs in one launch (csrc/s2c_graph.hip: s2c_local_feats / _grad) against
the framework ops they replace (models/caption_module.py:250-292, _add_relation_feat :394-414): values and
every gradient, with repeated local ids, ids that are no neighbour of the target, and without relation rows."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _torch_path(obj, rel, nbr, tgt, lid):
    B, K, F = obj.shape
    L = lid.shape[1]
    tf = torch.gather(obj, 1, tgt.view(B, 1, 1).expand(B, 1, F)).squeeze(1)
    o = obj
    if rel is not None:
        LR = rel.shape[2]
        r = torch.gather(rel, 1, tgt.view(B, 1, 1, 1).expand(B, 1, LR, F)).squeeze(1)
        n = torch.gather(nbr, 1, tgt.view(B, 1, 1).expand(B, 1, LR)).squeeze(1)
        o = obj.clone()
        o.scatter_add_(1, n.unsqueeze(-1).expand(B, LR, F), r)
    return tf, torch.gather(o, 1, lid.unsqueeze(-1).expand(B, L, F))


@pytest.mark.parametrize("B,K,L,F,with_rel", [(8, 256, 10, 128, True), (8, 256, 10, 128, False),
                                               (3, 40, 7, 32, True), (2, 16, 5, 200, True)])
def test_local_feats_match_the_framework_ops(B, K, L, F, with_rel):
    from scan2cap_amd.models.caption_module import _LocalFeats
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + K + L)
    obj = torch.randn(B, K, F, device="cuda", generator=g)
    rel = torch.randn(B, K, L, F, device="cuda", generator=g) if with_rel else None
    nbr = torch.stack([torch.stack([torch.randperm(K, device="cuda", generator=g)[:L].sort()[0]
                                    for _ in range(K)]) for _ in range(B)]) if with_rel else None
    tgt = torch.randint(0, K, (B,), device="cuda", generator=g)
    lid = torch.randint(0, K, (B, L), device="cuda", generator=g)
    lid[0, 1] = lid[0, 0]                                   # a repeated local id
    if with_rel:
        lid[0, 2] = nbr[0, tgt[0], 0]                       # a neighbour of the target: its relation row is added
        lid[1 % B, 0] = nbr[1 % B, tgt[1 % B], L - 1]
    w_tf = torch.randn(B, F, device="cuda", generator=g)
    w_lo = torch.randn(B, L, F, device="cuda", generator=g)
    res = []
    for fused in (True, False):
        o = obj.clone().requires_grad_(True)
        r = rel.clone().requires_grad_(True) if with_rel else None
        tf, lo = _LocalFeats.apply(o, r, nbr, tgt, lid) if fused else _torch_path(o, r, nbr, tgt, lid)
        ((tf * w_tf).sum() + (lo * w_lo).sum()).backward()
        res.append((tf.detach(), lo.detach(), o.grad, r.grad if with_rel else None))
    for a, b in zip(*res):
        if a is None:
            assert b is None
            continue
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())


def test_local_feats_only_the_local_gradient_arrives():
    """The target's features feed nothing (d_target None): the autograd engine hands the op one gradient."""
    from scan2cap_amd.models.caption_module import _LocalFeats
    torch.manual_seed(0)
    B, K, L, F = 4, 32, 6, 64
    obj = torch.randn(B, K, F, device="cuda", requires_grad=True)
    tgt = torch.randint(0, K, (B,), device="cuda")
    lid = torch.randint(0, K, (B, L), device="cuda")
    _, lo = _LocalFeats.apply(obj, None, None, tgt, lid)
    lo.sum().backward()
    want = torch.zeros(B, K, F, device="cuda")
    want.scatter_add_(1, lid.unsqueeze(-1).expand(B, L, F), torch.ones(B, L, F, device="cuda"))
    assert torch.equal(obj.grad, want)
