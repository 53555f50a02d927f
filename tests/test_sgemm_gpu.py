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


@pytest.mark.parametrize("R,T,V,H", [(8, 30, 3500, 512), (8, 13, 200, 128), (3, 7, 60, 64), (16, 30, 3500, 512)])
def test_classifier_products_with_row_maps(R, T, V, H):
    """The teacher-forced decoder's three classifier products on s2c_small_gemm_ex: rows (r, t) of the
    logits read H2[t + 1, r] in place (NT, two-level row map), dW_cls = dl^T H2 (TN: reduction over the
    R T rows), dH2[t, r] = dl W_cls (NN, K = V split over workgroups with float atomics)."""
    from scan2cap_amd import mgemm as mg
    torch.manual_seed(R + T + V)
    H2 = torch.randn(T + 1, R, H, device="cuda")
    W = torch.randn(V, H, device="cuda") * 0.1
    b = torch.randn(V, device="cuda")
    RH = R * H
    rows = mg.ax(RH, div=T, hi=H)
    logits = torch.empty(R, T, V, device="cuda")
    assert mg.mfma(mg.NT, R * T, V, H, H2[1:], rows, W, mg.ax(H), logits, mg.ax(V), bias=b)
    H2n = H2[1:].permute(1, 0, 2).contiguous().view(R * T, H)
    want = H2n.double() @ W.double().t() + b.double()
    assert ((logits.view(R * T, V).double() - want).abs().max() / want.abs().max()).item() < 2e-6
    dl = torch.randn(R * T, V, device="cuda")
    dW = torch.empty(V, H, device="cuda")
    assert mg.mfma(mg.TN, V, H, R * T, dl, mg.ax(V), H2[1:], rows, dW, mg.ax(H))
    want = dl.double().t() @ H2n.double()
    assert ((dW.double() - want).abs().max() / want.abs().max()).item() < 2e-6
    dH2 = torch.zeros(T, R, H, device="cuda")
    assert mg.mfma(mg.NN, R * T, H, V, dl, mg.ax(V), W, mg.ax(H), dH2, rows, ksplit=16)
    want = (dl.double() @ W.double()).view(R, T, H).permute(1, 0, 2)
    assert ((dH2.double() - want).abs().max() / want.abs().max()).item() < 2e-6


def test_ex_declines_what_it_cannot_address():
    from scan2cap_amd import mgemm as mg
    A = torch.randn(30, 62, device="cuda")            # V = 62: not a multiple of 4
    B = torch.randn(30, 64, device="cuda")
    Y = torch.empty(62, 64, device="cuda")
    assert not mg.mfma(mg.TN, 62, 64, 30, A, mg.ax(62), B, mg.ax(64), Y, mg.ax(64))
    assert not mg.mfma(mg.NN, 30, 62, 64, B, mg.ax(64), torch.randn(64, 62, device="cuda"), mg.ax(62),
                       torch.empty(30, 62, device="cuda"), mg.ax(62))
