"""Tall weight gradients on the streaming kernel (csrc/s2c_dwstream.hip) against a float64 product:
the shapes of the cfg3 step (lib/pointnet2/pytorch_utils.py:67-120 backward), ragged row counts,
operands read in place out of a wider tensor (the (B,N,3+C) cloud: N % 4 != 0, rows only
dword-aligned -- the DMA pieces of the tensor's last row must not be issued), the Gram matrix."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _run(dY, A):
    from scan2cap_amd.pointnet2 import fused
    pend = []
    dW = fused._weight_grad_stream(dY, A, pend)
    assert dW is not None, "shape not taken by the streaming kernel"
    fused.flush_partial_sums(pend)
    return dW


def _check(dY, A, tol=2e-6):
    got = _run(dY, A)
    want = dY.double().t() @ A.double()
    scale = (dY.double().norm(dim=0)[:, None] * A.double().norm(dim=0)[None, :]).clamp_min(1e-30)
    err = ((got.double() - want).abs() / scale).max().item()
    assert err < tol, err
    return got


@pytest.mark.parametrize("M,C,N", [
    (1 << 20, 64, 64), (262144, 128, 128), (262144, 256, 128), (65536, 256, 128),
    (320000, 64, 132), (40000, 64, 3), (65536, 128, 256), (131072, 64, 259),
    (33000, 64, 64), (32768 + 17, 128, 70), (100003, 192, 128), (50000, 64, 7), (40001, 64, 64)])
def test_weight_grad_stream_matches_float64(M, C, N):
    torch.manual_seed(M % 1000 + C + N)
    dY = torch.randn(M, C, device="cuda")
    A = torch.randn(M, N, device="cuda") * 0.7 + 0.1
    _check(dY, A)


def test_shapes_beyond_four_wave_tiles_are_declined():
    """More than eight tiles of 64 x 64, or C not a multiple of 64: the planner says no and
    _weight_grad falls through to its other kernels (the caller gets a correct dW either way)."""
    from scan2cap_amd.pointnet2 import fused
    for M, C, N in [(65536, 128, 259), (32768, 256, 256), (50000, 4, 7)]:
        dY = torch.randn(M, C, device="cuda")
        A = torch.randn(M, N, device="cuda")
        assert fused._weight_grad_stream(dY, A, []) is None
        pend = []
        dW = fused._weight_grad(dY, A, pend)
        fused.flush_partial_sums(pend)
        want = dY.double().t() @ A.double()
        assert ((dW.double() - want).abs().max() / want.abs().max()).item() < 1e-5


def test_operand_in_place_out_of_the_cloud():
    """A = the rows of a (B, n, 3 + C) cloud (135 floats: dword-aligned rows, N % 4 != 0) and its
    feature columns (offset 3, 132 columns); the storage ends with the last row."""
    torch.manual_seed(3)
    B, n, C = 8, 5000, 132
    cloud = torch.randn(B, n, 3 + C, device="cuda")
    Z = torch.randn(B * n, 64, device="cuda")
    rows = cloud.view(B * n, 3 + C)
    got = _check(Z, rows)
    feats = rows[:, 3:]
    assert feats.stride(0) == 135 and feats.data_ptr() == rows.data_ptr() + 12
    gf = _check(Z, feats)
    # same products, same chunking of the rows -> the shared columns agree to rounding of the sums
    assert torch.allclose(got[:, 3:], gf, rtol=0, atol=2e-3 * gf.abs().max().item())


def test_first_layer_weight_gradient_over_the_cloud_rows():
    """GatherSpec.weight_grad with xyz tagged as the cloud's first columns (the backbone does that):
    one product over the (B n, 3 + C) rows equals the two separate products."""
    from scan2cap_amd.pointnet2 import fused
    torch.manual_seed(11)
    B, n, C, m, ns, Cout = 8, 10000, 132, 256, 16, 64
    cloud = torch.randn(B, n, 3 + C, device="cuda")
    xyz = cloud[..., :3].contiguous()
    feats = cloud[..., 3:]
    new_xyz = xyz[:, :m].contiguous()
    idx = torch.randint(0, n, (B, m, ns), device="cuda", dtype=torch.int32)
    dY = torch.randn(B * m * ns, Cout, device="cuda")
    outs = []
    for tag in (False, True):
        x = xyz.clone()
        if tag:
            x._s2c_cloud = cloud
        g = fused.GatherSpec(x, new_xyz, feats, idx, 0.4, True)
        pend, post = [], []
        dW = g.weight_grad(dY, pending=pend, post=post)
        fused.flush_partial_sums(pend)
        for fin in post:
            fin()
        outs.append(dW)
    X = g.materialise()
    want = dY.double().t() @ X.double()
    for dW in outs:
        assert ((dW.double() - want).abs().max() / want.abs().max()).item() < 2e-5
    assert (outs[0] - outs[1]).abs().max() < 1e-3 * want.abs().max().item()


def test_gram_matrix_loads_the_operand_once():
    torch.manual_seed(5)
    A = torch.randn(1 << 18, 64, device="cuda").relu_()
    got = _check(A, A)
    assert torch.equal(got, got.t().contiguous()) or (got - got.t()).abs().max() < 1e-3 * got.abs().max()


def test_strided_left_operand_and_determinism():
    torch.manual_seed(7)
    wide = torch.randn(70000, 192, device="cuda")
    dY = wide[:, 64:128]            # ldy 192, 16-byte aligned column block
    A = torch.randn(70000, 96, device="cuda")
    a = _check(dY, A)
    b = _run(dY, A)
    assert torch.equal(a, b)


def test_default_weight_grad_takes_the_streaming_kernel_for_tall_layers():
    from scan2cap_amd.pointnet2 import fused
    dY = torch.randn(262144, 128, device="cuda")
    A = torch.randn(262144, 128, device="cuda")
    calls = []
    orig = fused._weight_grad_stream
    fused._weight_grad_stream = lambda *a: calls.append(1) or orig(*a)
    try:
        pend = []
        dW = fused._weight_grad(dY, A, pend)
        fused.flush_partial_sums(pend)
    finally:
        fused._weight_grad_stream = orig
    assert calls and (dW.double() - dY.double().t() @ A.double()).abs().max() < 1e-2


@pytest.mark.parametrize("M,C,N,relu", [(262144, 128, 128, 1), (65536, 256, 128, 1), (131072, 64, 64, 0),
                                        (32768 + 48, 128, 70, 1), (1 << 20, 64, 64, 1)])
def test_weight_grad_stream_with_the_activation_formed_on_the_way(M, C, N, relu):
    """dW = dY^T relu?(P scale + shift) (s2c_weight_grad_stream_act): the operand is the previous layer's
    pre-activation, the activation is never materialised; against float64 over the float32 activation."""
    from scan2cap_amd.pointnet2 import fused
    torch.manual_seed(M % 977 + C + N)
    dY = torch.randn(M, C, device="cuda")
    P = torch.randn(M, N, device="cuda") * 0.9 - 0.2
    scale = torch.randn(N, device="cuda") * 0.5 + 1.0
    shift = torch.randn(N, device="cuda") * 0.3
    scale[1] = -0.6
    pend = []
    dW = fused._weight_grad_stream(dY, P, pend, act=(scale, shift, relu))
    assert dW is not None
    fused.flush_partial_sums(pend)
    act = P * scale + shift
    if relu:
        act = torch.relu(act)
    want = dY.double().t() @ act.double()
    sc = (dY.double().norm(dim=0)[:, None] * act.double().norm(dim=0)[None, :]).clamp_min(1e-30)
    assert ((dW.double() - want).abs() / sc).max().item() < 2e-6


def test_stack_without_activation_side_outputs_matches_the_stack_that_keeps_them():
    """A 16 -> 128 -> 128 -> 256 stack over 262144 rows (SA2's shapes): with DW_STREAM_ACT the forward
    writes no activation side output for the layers whose weight gradient streams; every output and
    gradient equals the stack that keeps them."""
    import copy
    from scan2cap_amd.pointnet2 import fused
    torch.manual_seed(4)
    M = 262144
    convs = [torch.nn.Conv1d(a, b, 1, bias=False).cuda() for a, b in ((16, 128), (128, 128), (128, 256))]
    bns = [torch.nn.BatchNorm1d(c).cuda() for c in (128, 128, 256)]
    x = torch.randn(M, 16, device="cuda")
    state = copy.deepcopy([bn.state_dict() for bn in bns])

    def run(pool_ns, gout):
        for bn, st in zip(bns, state):
            bn.load_state_dict(st)
        specs = [fused.LayerSpec(False, bn, True) for bn in bns]
        params = []
        for c, bn in zip(convs, bns):
            params += [c.weight.view(c.weight.shape[0], -1), bn.weight, bn.bias]
            c.weight.grad = bn.weight.grad = bn.bias.grad = None
        xx = x.clone().requires_grad_(True)
        out = fused.mlp_rows(xx, specs, params, pool_ns=pool_ns)
        (out * gout).sum().backward()
        return [out.detach(), xx.grad] + [c.weight.grad.clone() for c in convs] + \
            [b.weight.grad.clone() for b in bns] + [b.bias.grad.clone() for b in bns]

    for pool_ns in (0, 32):
        gout = torch.randn(M // pool_ns if pool_ns else M, 256, device="cuda")
        seen = []
        real = fused._weight_grad_stream

        def spy(dY, A, pending, act=None):
            seen.append(act is not None)
            return real(dY, A, pending, act=act)
        fused._weight_grad_stream = spy
        try:
            new = run(pool_ns, gout)
        finally:
            fused._weight_grad_stream = real
        assert any(seen), "no weight gradient took the activation-on-the-way kernel"
        old_flag = fused.DW_STREAM_ACT
        fused.DW_STREAM_ACT = False
        try:
            old = run(pool_ns, gout)
        finally:
            fused.DW_STREAM_ACT = old_flag
        for a, b in zip(new, old):
            assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() < 2e-5
