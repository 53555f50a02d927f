"""scan2cap_amd.optim.FusedAdam (csrc/s2c_optim.hip: every parameter tensor's Adam update in one
launch) against torch.optim.Adam -- the reference's optimizer (scripts/train.py:138, stepped by
lib/solver.py:293-302)."""
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

SHAPES = [(64, 135), (64,), (64,), (3,), (1, 1), (128, 64, 1, 1), (259, 256, 1), (97,), (3500, 512),
          (4099,), (1536, 812), (7, 5, 3)]


def _params(seed, device):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(device)) for s in SHAPES]


def _grads(params, seed, skip=()):
    g = torch.Generator().manual_seed(seed)
    for i, p in enumerate(params):
        gr = torch.randn(tuple(p.shape), generator=g) * (10.0 ** ((i % 5) - 3))
        p.grad = None if i in skip else gr.to(device=p.device, dtype=p.dtype)


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(1e-12, float(b.double().abs().max())))


@pytest.mark.parametrize("wd", [0.0, 1e-5, 1e-2])
def test_fused_adam_tracks_torch_adam(wd):
    """Ten updates with changing gradients; parameter 3 gets no gradient in steps 2 and 5 (torch skips
    it and keeps its step count); every parameter and both moments within 2e-6 of torch.optim.Adam
    run in float64, and closer to it than 3 x torch's own float32 step is."""
    from scan2cap_amd.optim import FusedAdam
    dev = torch.device("cuda")
    mine, ref32 = _params(1, dev), _params(1, dev)
    ref64 = [torch.nn.Parameter(p.detach().double().cpu()) for p in _params(1, "cpu")]
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    o_mine, o32, o64 = FusedAdam(mine, **kw), torch.optim.Adam(ref32, **kw), torch.optim.Adam(ref64, **kw)
    for it in range(10):
        skip = (3,) if it in (2, 5) else ()
        _grads(mine, 100 + it, skip)
        _grads(ref32, 100 + it, skip)
        _grads(ref64, 100 + it, skip)
        o_mine.step()
        o32.step()
        o64.step()
    assert o_mine._fallback is False
    for i, (a, b, c) in enumerate(zip(mine, ref32, ref64)):
        e_mine, e_ref = _rel(a, c), _rel(b, c)
        assert e_mine <= max(2e-6, 3 * e_ref), (i, e_mine, e_ref)
        st, st64 = o_mine.state[a], o64.state[c]
        assert float(st["step"]) == float(st64["step"]), i
        assert _rel(st["exp_avg"], st64["exp_avg"]) <= 2e-6, i
        assert _rel(st["exp_avg_sq"], st64["exp_avg_sq"]) <= 2e-6, i


def test_fused_adam_state_dict_round_trips_with_torch_adam(tmp_path):
    """lib/solver.py:501-515 / scripts/train.py:138-145: a checkpoint written by one optimizer resumes in
    the other, in place (scan2cap_amd.checkpoint.load_optimizer_state_inplace) and through
    `load_state_dict`, and the next update agrees."""
    from scan2cap_amd import checkpoint
    from scan2cap_amd.optim import FusedAdam
    dev = torch.device("cuda")
    kw = dict(lr=2e-3, weight_decay=1e-5)
    a, b = _params(2, dev), _params(2, dev)
    oa, ob = FusedAdam(a, **kw), torch.optim.Adam(b, capturable=True, **kw)
    for it in range(3):
        _grads(a, it)
        _grads(b, it)
        oa.step()
        ob.step()
    # torch -> fused (in place and by load_state_dict), fused -> torch
    c, d, e = _params(2, dev), _params(2, dev), _params(2, dev)
    oc, od = FusedAdam(c, **kw), FusedAdam(d, **kw)
    oe = torch.optim.Adam(e, capturable=True, **kw)
    for q, src in ((c, b), (d, b), (e, a)):
        for x, y in zip(q, src):
            x.data.copy_(y.data)
    _grads(c, 50)
    oc.step()                      # populated state first: the in-place path overwrites live tensors
    for x, y in zip(c, b):
        x.data.copy_(y.data)
    checkpoint.load_optimizer_state_inplace(oc, ob.state_dict())
    # (a deep copy, as a checkpoint read from disk is: Optimizer.load_state_dict keeps the tensors it is
    # given when dtype and device already fit, and `ob` goes on stepping below)
    od.load_state_dict(copy.deepcopy(ob.state_dict()))
    torch.save(oa.state_dict(), tmp_path / "opt.pt")
    oe.load_state_dict(torch.load(tmp_path / "opt.pt"))
    for q in (a, b, c, d, e):
        _grads(q, 77)
    for o in (oa, ob, oc, od, oe):
        o.step()
    assert float(oc.state[c[0]]["step"]) == 4.0 and float(od.state[d[0]]["step"]) == 4.0
    assert float(oe.state[e[0]]["step"]) == 4.0 and float(oa.state[a[0]]["step"]) == 4.0
    for name, q in (("fused all along", a), ("torch -> fused, in place", c),
                    ("torch -> fused, load_state_dict", d), ("fused -> torch", e)):
        for i in range(len(SHAPES)):
            assert _rel(q[i], b[i]) <= 2e-6, (name, i, _rel(q[i], b[i]))


def test_fused_adam_inside_a_captured_graph():
    """The step count lives on the device: replays of a captured step advance it."""
    from scan2cap_amd.graphs import GraphedCallable
    from scan2cap_amd.optim import FusedAdam
    dev = torch.device("cuda")
    a, b = _params(3, dev), _params(3, dev)
    oa, ob = FusedAdam(a, lr=1e-3, weight_decay=1e-5), torch.optim.Adam(b, lr=1e-3, weight_decay=1e-5)
    _grads(a, 5)
    _grads(b, 5)
    start = [p.detach().clone() for p in a]
    g = GraphedCallable(lambda: oa.step(), warmup=2).capture()     # 2 warm-up steps + the capture pass
    for p, s in zip(a, start):
        p.data.copy_(s)
    for st in oa.state.values():
        for v in st.values():
            v.zero_()
    for _ in range(6):
        g()
        ob.step()
    torch.cuda.synchronize()
    assert float(oa.state[a[0]]["step"]) == 6.0
    for i, (x, y) in enumerate(zip(a, b)):
        assert _rel(x, y) <= 5e-6, i


def test_fused_adam_many_tensors_and_two_groups():
    """More tensors than one launch's argument block holds (128) and two parameter groups with their
    own learning rate / weight decay: still torch.optim.Adam's numbers."""
    from scan2cap_amd.optim import FusedAdam
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    shapes = [(1 + (i * 37) % 300,) for i in range(150)] + [(70, 33), (5000,)]

    def make():
        gg = torch.Generator().manual_seed(9)
        return [torch.nn.Parameter(torch.randn(s, generator=gg).to(dev)) for s in shapes]
    a, b = make(), make()
    groups = lambda ps: [dict(params=ps[:100], lr=1e-3, weight_decay=1e-5),
                         dict(params=ps[100:], lr=5e-4, weight_decay=0.0)]
    oa, ob = FusedAdam(groups(a)), torch.optim.Adam(groups(b))
    for it in range(4):
        for x, y in zip(a, b):
            gr = torch.randn(tuple(x.shape), generator=g)
            x.grad, y.grad = gr.to(dev), gr.to(dev).clone()
        oa.step()
        ob.step()
    assert oa._fallback is False
    for i, (x, y) in enumerate(zip(a, b)):
        assert _rel(x, y) <= 5e-6, i
        assert float(oa.state[x]["step"]) == 4.0
