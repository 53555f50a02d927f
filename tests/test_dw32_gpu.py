"""csrc/s2c_dw32.hip (tall weight gradients on the fp32 matrix cores, dense and gather-fused operand)
against float64 products and against the materialised gathered rows (s2c_sa_gather_rows)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("M,C,K,ldx", [(40000, 64, 64, 64), (33001, 128, 131, 135), (32768, 256, 128, 128),
                                       (5000, 64, 3, 3), (70000, 128, 259, 259), (1234, 97, 128, 128)])
def test_weight_grad_f32_dense_matches_float64(M, C, K, ldx):
    from scan2cap_amd.pointnet2 import fused
    g = torch.Generator(device="cuda").manual_seed(M)
    dY = torch.randn(M, C, device="cuda", generator=g)
    Xs = torch.randn(M, ldx, device="cuda", generator=g)
    X = Xs[:, :K]
    pending = []
    dW = fused._weight_grad_f32(dY, X, pending)
    assert dW is not None and len(pending) == 1
    fused.flush_partial_sums(pending)
    want = dY.double().t() @ X.double()
    assert _rel(dW, want) < 2e-6


@pytest.mark.parametrize("B,N,m,ns,Cf,normalize", [(2, 3000, 256, 32, 128, True), (3, 1000, 100, 16, 256, False),
                                                   (1, 500, 64, 64, 0, True), (2, 2048, 1024, 32, 61, True)])
def test_weight_grad_f32_reads_the_gathered_operand_in_place(B, N, m, ns, Cf, normalize):
    from scan2cap_amd.pointnet2 import fused
    g = torch.Generator(device="cuda").manual_seed(B * N + m)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g) * 4 - 2
    new_xyz = xyz[:, :m].contiguous()
    feats = torch.randn(B, N, Cf + 5, device="cuda", generator=g)[..., 2:2 + Cf] if Cf else None
    idx = torch.randint(0, N, (B, m, ns), device="cuda", generator=g, dtype=torch.int32)
    spec = fused.GatherSpec(xyz, new_xyz, feats, idx, 0.4, normalize)
    G = spec.materialise()                                     # (rows, 3 + Cf)
    dY = torch.randn(spec.rows, 96, device="cuda", generator=g)
    pending = []
    dW = fused._weight_grad_f32(dY, None, pending, gather=spec)
    assert dW is not None
    fused.flush_partial_sums(pending)
    assert _rel(dW, dY.double().t() @ G.double()) < 2e-6
