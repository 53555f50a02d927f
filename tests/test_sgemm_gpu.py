"""The small products of the layer stacks on csrc/s2c_sgemm.hip against float64: dX = dY W with W as
stored, y = x W^T + b; ragged M / N / K (K = 259 and 97 occur in the cfg3 step), strided operands."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _err(got, want, A, B, transposed):
    Bd = B.double().t() if transposed else B.double()
    scale = (A.double().norm(dim=1)[:, None] * Bd.norm(dim=0)[None, :]).clamp_min(1e-30)
    return ((got.double() - want).abs() / scale).max().item()


@pytest.mark.parametrize("M,K,N", [
    (8192, 256, 256), (20480, 128, 128), (20480, 128, 256), (8192, 256, 512), (8192, 259, 256),
    (4096, 256, 512), (2048, 128, 128), (2048, 97, 128), (4096, 256, 256), (100, 64, 64),
    (777, 300, 132), (64, 4, 4), (1, 130, 68), (4097, 513, 260)])
@pytest.mark.parametrize("transposed", [False, True])
def test_small_gemm_matches_float64(M, K, N, transposed):
    from scan2cap_amd.pointnet2 import fused
    torch.manual_seed(M + K + N)
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") if transposed else torch.randn(K, N, device="cuda")
    bias = torch.randn(N, device="cuda") if transposed else None
    Y = fused.small_gemm(A, B, transposed, bias)
    assert Y is not None
    want = A.double() @ (B.double().t() if transposed else B.double())
    if bias is not None:
        want = want + bias.double()
    assert _err(Y, want, A, B, transposed) < 2e-6


def test_small_gemm_strided_operands_and_nan_containment():
    """A column block of a wider tensor as the operand; a NaN in a row of A stays in that row of Y
    (the ragged-K fix-up must not leak a neighbouring row's values into a product)."""
    from scan2cap_amd.pointnet2 import fused
    torch.manual_seed(1)
    wide = torch.randn(3000, 400, device="cuda")
    A = wide[:, 3:262]                       # K = 259, rows 12 bytes off a 16-byte boundary
    W = torch.randn(259, 256, device="cuda")
    wide[1234, 3] = float("nan")
    Y = fused.small_gemm(A, W, False)
    assert Y is not None
    bad = torch.isnan(Y).any(dim=1)
    assert bad[1234] and int(bad.sum()) == 1
    ok = ~bad
    want = A.double()[ok] @ W.double()
    assert ((Y.double()[ok] - want).abs().max() / want.abs().max()).item() < 1e-5


def test_unsupported_layouts_are_declined():
    from scan2cap_amd.pointnet2 import fused
    A = torch.randn(512, 64, device="cuda")
    assert fused.small_gemm(A, torch.randn(64, 130, device="cuda"), False) is None      # N % 4
    assert fused.small_gemm(A, torch.randn(130, 64, device="cuda"), True) is not None   # any N in the W^T form
    assert fused.small_gemm(A[:, :2], torch.randn(2, 64, device="cuda"), False) is None  # K < 4
