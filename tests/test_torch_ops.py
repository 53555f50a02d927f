"""`torch.ops.s2c.*` (scan2cap_amd/pointnet2/ops.py): the nine operators as registered
PyTorch custom ops on top of the C ABI -- schema, fake (meta) kernels, autograd, and no
CPU kernel (bindings.cpp:6-19 / "CPU not supported")."""
import numpy as np
import pytest
import torch

from scan2cap_amd.pointnet2 import ops  # noqa: F401  (registration)
from scan2cap_amd.synthetic import scene_xyz


def test_all_nine_ops_are_registered_with_schemas():
    for name in ops.NAMES:
        op = getattr(torch.ops.s2c, name)
        assert op.default._schema.name == "s2c::" + name
    s = str(torch.ops.s2c.ball_query.default._schema)
    assert s == "s2c::ball_query(Tensor new_xyz, Tensor xyz, float radius, int nsample) -> Tensor"


def test_fake_kernels_infer_shapes_without_a_gpu():
    B, N, m, ns, C = 2, 100, 16, 8, 5
    with torch.device("meta"):
        xyz = torch.empty(B, N, 3)
        new_xyz = torch.empty(B, m, 3)
        feats = torch.empty(B, C, N)
        idx1 = torch.empty(B, m, dtype=torch.int32)
        idx2 = torch.empty(B, m, ns, dtype=torch.int32)
        idx3 = torch.empty(B, N, 3, dtype=torch.int32)
        w = torch.empty(B, N, 3)
    o = torch.ops.s2c
    assert o.furthest_point_sampling(xyz, m).shape == (B, m)
    assert o.furthest_point_sampling(xyz, m).dtype == torch.int32
    assert o.gather_points(feats, idx1).shape == (B, C, m)
    assert o.gather_points_grad(o.gather_points(feats, idx1), idx1, N).shape == (B, C, N)
    assert o.ball_query(new_xyz, xyz, 0.2, ns).shape == (B, m, ns)
    assert o.group_points(feats, idx2).shape == (B, C, m, ns)
    assert o.group_points_grad(o.group_points(feats, idx2), idx2, N).shape == (B, C, N)
    d2, i3 = o.three_nn(xyz, new_xyz)
    assert d2.shape == (B, N, 3) and i3.shape == (B, N, 3) and i3.dtype == torch.int32
    known = torch.empty(B, C, m, device="meta")
    assert o.three_interpolate(known, idx3, w).shape == (B, C, N)
    assert o.three_interpolate_grad(o.three_interpolate(known, idx3, w), idx3, w, m).shape \
        == (B, C, m)


def test_there_is_no_cpu_kernel():
    with pytest.raises(NotImplementedError):
        torch.ops.s2c.furthest_point_sampling(torch.zeros(1, 8, 3), 4)


@pytest.mark.gpu
def test_ops_match_the_c_abi_shim_and_differentiate(oracle):
    from scan2cap_amd.pointnet2 import _ext
    dev = "cuda"
    B, N, m, ns, C = 2, 5000, 256, 16, 7
    xyz_np = scene_xyz(B, N, seed=3, mode="surface")
    xyz = torch.from_numpy(xyz_np).to(dev)
    o = torch.ops.s2c
    inds = o.furthest_point_sampling(xyz, m)
    assert torch.equal(inds, _ext.furthest_point_sampling(xyz, m))
    assert np.array_equal(inds.cpu().numpy(), oracle.furthest_point_sampling(xyz_np, m))
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = o.ball_query(new_xyz, xyz, 0.3, ns)
    assert np.array_equal(idx.cpu().numpy(),
                          oracle.ball_query(new_xyz.cpu().numpy(), xyz_np, 0.3, ns))
    feats = torch.randn(B, C, N, device=dev, requires_grad=True)
    # autograd registered on the ops themselves (no autograd.Function wrapper needed)
    g = o.group_points(feats, idx)
    w = torch.randn_like(g)
    (g * w).sum().backward()
    want = oracle.group_points_grad(w.cpu().numpy(), idx.cpu().numpy(), N)
    np.testing.assert_allclose(feats.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    feats.grad = None
    ga = o.gather_points(feats, inds)
    w = torch.randn_like(ga)
    (ga * w).sum().backward()
    np.testing.assert_allclose(
        feats.grad.cpu().numpy(),
        oracle.gather_points_grad(w.cpu().numpy(), inds.cpu().numpy(), N), rtol=1e-5, atol=1e-5)
    d2, i3 = o.three_nn(xyz[:, :700].contiguous(), new_xyz)
    kn = torch.randn(B, C, m, device=dev, requires_grad=True)
    wt = torch.rand(B, 700, 3, device=dev)
    out = o.three_interpolate(kn, i3, wt)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    np.testing.assert_allclose(
        kn.grad.cpu().numpy(),
        oracle.three_interpolate_grad(w.cpu().numpy(), i3.cpu().numpy(), wt.cpu().numpy(), m),
        rtol=1e-4, atol=1e-5)
    # the library's own consistency checker (schema, fake kernel, autograd registration)
    torch.library.opcheck(o.group_points.default, (feats.detach().requires_grad_(True), idx),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
