"""A 64 -> 64 BatchNorm(+ReLU) layer's backward in one pass (csrc/s2c_bnbwd_fused.hip: dY never written,
dX = dY W, dW = dY^T act(previous layer), the previous layer's column sums) against float64 formulas
and against the three-launch path it replaces (s2c_bn_bwd_gemm_next_stats + dY + s2c_weight_grad_stream).
Reference: autograd of lib/pointnet2/pytorch_utils.py:67-120 inside pointnet2_modules.py:251-257."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _inputs(M, seed, relu, nrelu):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    d = dict(dA=r(M, 64), Y=r(M, 64) * 1.3 + 0.2, nY=r(M, 64) * 0.8 - 0.1, W=r(64, 64) * 0.2,
             scale=r(64) * 0.5 + 1.0, shift=r(64) * 0.3, mean=r(64) * 0.2,
             invstd=r(64).abs() + 0.5, coef=torch.cat([r(64) * 0.5 + 1.0, r(64) * 0.01, r(64) * 0.01]),
             nscale=r(64) * 0.5 + 1.0, nshift=r(64) * 0.3, nmean=r(64) * 0.2, ninvstd=r(64).abs() + 0.5)
    d["scale"][3] = -0.7            # a negative BatchNorm weight flips the ReLU mask's sense
    d["relu"], d["nrelu"] = int(relu), int(nrelu)
    return d


def _run(d):
    from scan2cap_amd.pointnet2 import fused
    M = d["dA"].shape[0]
    parts = fused._bwd_dx_dw_parts(M)
    assert parts > 0
    dX = torch.full((M, 64), float("nan"), device="cuda")
    wpart = torch.full((parts, 64, 64), float("nan"), device="cuda")
    npart = torch.full((parts, 128), float("nan"), device="cuda")
    fused._call("s2c_bn_bwd_dx_dw64", dX, M, d["dA"].data_ptr(), d["Y"].data_ptr(),
                d["scale"].data_ptr(), d["shift"].data_ptr(), d["mean"].data_ptr(),
                d["invstd"].data_ptr(), d["coef"].data_ptr(), d["relu"], d["W"].data_ptr(),
                d["W"].stride(0), dX.data_ptr(), d["nY"].data_ptr(), d["nscale"].data_ptr(),
                d["nshift"].data_ptr(), d["nmean"].data_ptr(), d["ninvstd"].data_ptr(), d["nrelu"],
                wpart.data_ptr(), npart.data_ptr())
    return dX, wpart.sum(0), npart.sum(0)


def _truth(d):
    """float64 formulas on the float32 masks (the masks are comparisons of float32 expressions:
    a float64 mask would differ on knife-edge elements)."""
    f = lambda k: d[k].double()
    mask = (d["Y"] * d["scale"] + d["shift"] > 0) if d["relu"] else torch.ones_like(d["Y"], dtype=torch.bool)
    dz = torch.where(mask, f("dA"), torch.zeros((), dtype=torch.float64, device="cuda"))
    k0, k1, k2 = f("coef")[:64], f("coef")[64:128], f("coef")[128:]
    dY = k0 * (dz - k1 - ((f("Y") - f("mean")) * f("invstd")) * k2)
    dX = dY @ f("W")
    pre = d["nY"] * d["nscale"] + d["nshift"]                      # float32, as the kernels form it
    act = (torch.relu(pre) if d["nrelu"] else pre).double()
    dW = dY.t() @ act
    nmask = (pre > 0) if d["nrelu"] else torch.ones_like(pre, dtype=torch.bool)
    # the column sums are over the float32 dX the kernel wrote; its own rounding is checked through dX
    return dY, dX, dW, nmask


@pytest.mark.parametrize("M,relu,nrelu", [(1 << 20, 1, 1), (65536, 1, 1), (4096, 1, 0), (40000 - 40000 % 16, 0, 1),
                                          (16 * 12345, 1, 1)])
def test_fused_layer_backward_matches_float64(M, relu, nrelu):
    d = _inputs(M, 11 + M % 97, relu, nrelu)
    dX, dW, st = _run(d)
    dY, tX, tW, nmask = _truth(d)
    sx = (dY.norm(dim=1)[:, None] * d["W"].double().norm(dim=0)[None, :]).clamp_min(1e-30)
    ex = ((dX.double() - tX).abs() / sx).max().item()
    assert ex < 2e-6, ex
    pre = d["nY"] * d["nscale"] + d["nshift"]
    act = (torch.relu(pre) if nrelu else pre).double()
    sw = (dY.norm(dim=0)[:, None] * act.norm(dim=0)[None, :]).clamp_min(1e-30)
    ew = ((dW.double() - tW).abs() / sw).max().item()
    assert ew < 2e-6, ew
    dz = torch.where(nmask, dX.double(), torch.zeros((), dtype=torch.float64, device="cuda"))
    xhat = (d["nY"].double() - d["nmean"].double()) * d["ninvstd"].double()
    t1, t2 = dz.sum(0), (dz * xhat).sum(0)
    n1, n2 = dz.abs().sum(0).clamp_min(1e-30), (dz * xhat).abs().sum(0).clamp_min(1e-30)
    assert ((st[:64].double() - t1).abs() / n1).max().item() < 2e-6
    assert ((st[64:].double() - t2).abs() / n2).max().item() < 2e-6


def test_fused_layer_backward_agrees_with_the_three_launch_path():
    """The same layer through s2c_bn_bwd_gemm_next_stats (dY written) + the streaming weight gradient over
    (dY, the activation tensor): dX to rounding, dW and the column sums to their summation order."""
    from scan2cap_amd.pointnet2 import fused
    M = 131072
    d = _inputs(M, 5, 1, 1)
    dX, dW, st = _run(d)
    Wt = d["W"].t().contiguous()
    dY = torch.empty(M, 64, device="cuda")
    dX2 = torch.empty(M, 64, device="cuda")
    nbg = fused._gemm_blocks(M, 64)
    npart = torch.empty(nbg * 128, device="cuda")
    fused._call("s2c_bn_bwd_gemm_next_stats", dX2, M, 64, 64, d["dA"].data_ptr(), d["Y"].data_ptr(),
                d["scale"].data_ptr(), d["shift"].data_ptr(), d["mean"].data_ptr(),
                d["invstd"].data_ptr(), d["coef"].data_ptr(), 1, Wt.data_ptr(), Wt.stride(0),
                dY.data_ptr(), dX2.data_ptr(), 64, d["nY"].data_ptr(), d["nscale"].data_ptr(),
                d["nshift"].data_ptr(), d["nmean"].data_ptr(), d["ninvstd"].data_ptr(), 1,
                npart.data_ptr())
    act = torch.relu(d["nY"] * d["nscale"] + d["nshift"])
    pend = []
    dW2 = fused._weight_grad_stream(dY, act, pend)
    fused.flush_partial_sums(pend)
    assert ((dX - dX2).abs().max() / dX2.abs().max()).item() < 2e-6
    assert ((dW - dW2).abs().max() / dW2.abs().max()).item() < 2e-6
    st2 = npart.view(nbg, 128).sum(0)
    assert ((st - st2).abs().max() / st2.abs().max()).item() < 2e-5


def test_fused_layer_backward_is_deterministic_and_declines_ragged_rows():
    from scan2cap_amd.pointnet2 import fused
    d = _inputs(65536, 3, 1, 1)
    a = _run(d)
    b = _run(d)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert fused._bwd_dx_dw_parts(65536 + 8) == 0
    assert fused._bwd_dx_dw_parts(1024) == 0


def _stack(seed=0, C0=16):
    torch.manual_seed(seed)
    convs = [torch.nn.Conv1d(a, b, 1, bias=False).cuda() for a, b in ((C0, 64), (64, 64), (64, 128))]
    bns = [torch.nn.BatchNorm1d(c).cuda() for c in (64, 64, 128)]
    for bn in bns:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    with torch.no_grad():
        bns[1].weight[5] = -0.8
    return convs, bns


def _stack_grads(convs, bns, x, gout, pool_ns):
    from scan2cap_amd.pointnet2 import fused
    specs = [fused.LayerSpec(False, bn, True) for bn in bns]
    params = []
    for c, bn in zip(convs, bns):
        params += [c.weight.view(c.weight.shape[0], -1), bn.weight, bn.bias]
    for c, bn in zip(convs, bns):
        c.weight.grad = bn.weight.grad = bn.bias.grad = None
    xx = x.detach().clone().requires_grad_(True)
    out = fused.mlp_rows(xx, specs, params, pool_ns=pool_ns)
    (out * gout).sum().backward()
    return out.detach(), xx.grad, [c.weight.grad.clone() for c in convs], \
        [bn.weight.grad.clone() for bn in bns], [bn.bias.grad.clone() for bn in bns]


def _flat(res):
    out, gx, gw, gg, gb = res
    return [out, gx] + gw + gg + gb


@pytest.mark.parametrize("M,pool_ns", [(65536, 0), (262144, 64), (262144, 0)])
def test_stack_with_the_one_pass_layer_matches_the_three_launch_stack(M, pool_ns):
    """A set-abstraction-like stack 16 -> 64 -> 64 -> 128: its middle layer runs on s2c_bn_bwd_dx_dw64 (no
    dY tensor; from 131072 rows on -- the streaming forward GEMM -- no activation side output in the forward
    either); every gradient against the same stack with FUSE_BWD_DX_DW off (which the stack tests of
    tests/test_fused_gpu.py hold against torch; the kernel itself is held against float64 above -- a float64
    run of the whole stack flips ReLU masks on knife-edge elements and is no reference for per-row gradients)."""
    import copy
    from scan2cap_amd.pointnet2 import fused
    convs, bns = _stack()
    x = torch.randn(M, 16, device="cuda")
    rows = M // pool_ns if pool_ns else M
    gout = torch.randn(rows, 128, device="cuda")
    state = copy.deepcopy([bn.state_dict() for bn in bns])
    calls = []
    real = fused._call

    def spy(name, *a, **k):
        calls.append(name)
        return real(name, *a, **k)
    fused._call = spy
    try:
        new = _flat(_stack_grads(convs, bns, x, gout, pool_ns))
    finally:
        fused._call = real
    # (the LAST layer, 64 -> 128, keeps the GEMM with the dY side output when it is not pooled)
    assert calls.count("s2c_bn_bwd_dx_dw64") == 1
    assert calls.count("s2c_bn_bwd_gemm_next_stats") == (0 if pool_ns else 1)
    for bn, st in zip(bns, state):
        bn.load_state_dict(st)
    old_flag = fused.FUSE_BWD_DX_DW
    fused.FUSE_BWD_DX_DW = False
    try:
        old = _flat(_stack_grads(convs, bns, x, gout, pool_ns))
    finally:
        fused.FUSE_BWD_DX_DW = old_flag
    for a, b in zip(new, old):
        assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() < 2e-5


def test_activation_not_kept_is_recomputed_when_the_backward_takes_another_path():
    """The forward skips the activation side output expecting the one-pass backward; if the products are
    switched to the exact fp32 chain in between, the backward materialises the activation itself."""
    import copy
    from scan2cap_amd.pointnet2 import fused
    M = 65536
    convs, bns = _stack(3)
    x = torch.randn(M, 16, device="cuda")
    gout = torch.randn(M, 128, device="cuda")
    state = copy.deepcopy([bn.state_dict() for bn in bns])
    ref = _flat(_stack_grads(convs, bns, x, gout, 0))
    for bn, st in zip(bns, state):
        bn.load_state_dict(st)
    specs = [fused.LayerSpec(False, bn, True) for bn in bns]
    params = []
    for c, bn in zip(convs, bns):
        params += [c.weight.view(c.weight.shape[0], -1), bn.weight, bn.bias]
        c.weight.grad = bn.weight.grad = bn.bias.grad = None
    xx = x.clone().requires_grad_(True)
    out = fused.mlp_rows(xx, specs, params)
    old_flag = fused.FUSE_BWD_DX_DW
    fused.FUSE_BWD_DX_DW = False
    try:
        (out * gout).sum().backward()
    finally:
        fused.FUSE_BWD_DX_DW = old_flag
    got = [out.detach(), xx.grad] + [c.weight.grad for c in convs] + [b.weight.grad for b in bns] \
        + [b.bias.grad for b in bns]
    for a, b in zip(got, ref):
        assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() < 2e-5
