"""The first layer of a set-abstraction stack in point space (csrc/s2c_sa.hip: sa_gather_add,
pointnet2/fused.py POINT_SPACE): Y = P[idx] + W_x rel against a float64 evaluation of
QueryAndGroup + the first 1x1 convolution (pointnet2_utils.py:347-359, pytorch_utils.py:67-120),
its BatchNorm column sums, and the whole stack (forward, weight / feature / coordinate gradients)
against the round-3 gather GEMM path."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("B,n,m,ns,C,N,normalize", [
    (2, 3000, 256, 32, 128, 128, True),      # SA2-like
    (8, 5000, 260, 64, 132, 64, True),       # SA1-like: 133120 rows -> 256-row workgroups
    (3, 700, 50, 16, 0, 64, False),          # no features: the relative position only
    (1, 500, 33, 7, 20, 256, True),          # ragged: 231 rows, 64 lanes per row
    (2, 900, 61, 16, 64, 36, True),          # 36 columns: 9 of 16 lanes per row
])
def test_gather_add_matches_float64_and_leaves_the_column_sums(B, n, m, ns, C, N, normalize):
    from scan2cap_amd import _C
    from scan2cap_amd.pointnet2 import fused
    g = torch.Generator(device="cuda").manual_seed(B * n + N)
    xyz = torch.rand(B, n, 3, device="cuda", generator=g) * 6 - 3
    new_xyz = xyz[:, :m].contiguous()
    feats = torch.randn(B, n, C, device="cuda", generator=g) if C else None
    idx = torch.randint(0, n, (B, m, ns), device="cuda", generator=g, dtype=torch.int32)
    W = torch.randn(N, 3 + C, device="cuda", generator=g) / (3 + C) ** 0.5
    radius = 0.4
    rows = B * m * ns
    P = None
    if C:
        P = torch.empty(B * n, N, device="cuda")
        fused._call("s2c_rows_gemm", P, B * n, N, C, feats.data_ptr(), C, W[:, 3:].data_ptr(),
                    W.stride(0), None, None, P.data_ptr(), N, None)
    nb = fused._gather_add_blocks(rows)
    part = torch.full((nb * 2 * N,), float("nan"), device="cuda")
    Y = torch.full((rows, N), float("nan"), device="cuda")
    fused._call("s2c_sa_gather_add", Y, B, n, m, ns, N, radius, int(normalize), xyz.data_ptr(),
                new_xyz.data_ptr(), fused._ptr(P), idx.data_ptr(), W.data_ptr(), W.stride(0),
                Y.data_ptr(), part.data_ptr())
    bi = torch.arange(B, device="cuda").view(B, 1, 1).expand(B, m, ns)
    rel = xyz[bi, idx.long()] - new_xyz.unsqueeze(2)                  # fp32, as the reference
    if normalize:
        rel = rel / radius
    X = rel.double()
    if C:
        X = torch.cat([X, feats[bi, idx.long()].double()], -1)
    want = (X.view(rows, -1) @ W.double().t())
    assert _rel(Y, want) < 2e-6
    part = part.view(nb, 2, N).double().sum(0)
    assert _rel(part[0], Y.double().sum(0)) < 1e-6
    assert _rel(part[1], (Y.double() ** 2).sum(0)) < 1e-6


@pytest.mark.parametrize("input_grad", [True, False])
@pytest.mark.parametrize("B,N,C,npoint,radius,ns,mlp", [
    (2, 2048, 128, 1024, 0.4, 32, [128, 128, 256]),
    (2, 4096, 132, 512, 0.2, 64, [64, 64, 128]),
])
def test_point_space_stack_matches_the_gather_gemm_stack(input_grad, B, N, C, npoint, radius, ns, mlp):
    from scan2cap_amd.pointnet2 import fused
    from scan2cap_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from scan2cap_amd.synthetic import scene_xyz
    torch.manual_seed(1)
    sa = PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=ns, mlp=[C] + mlp,
                               use_xyz=True, normalize_xyz=True).cuda().train()
    ref = copy.deepcopy(sa)
    cloud = torch.cat([torch.from_numpy(scene_xyz(B, N, seed=4)).cuda(),
                       torch.randn(B, N, C, device="cuda")], -1)      # rows of 3 + C floats
    outs = []
    for mod, on in ((sa, True), (ref, False)):
        fused.POINT_SPACE = on
        try:
            pc = cloud.clone()
            xyz = pc[..., :3].contiguous().requires_grad_(input_grad)
            feats = pc[..., 3:].requires_grad_(input_grad)            # a strided view, like SA1's
            nx, nf, ni = mod(xyz, feats.transpose(1, 2))
            gsum = (nf * torch.linspace(-1, 1, nf.numel(), device="cuda").view_as(nf)).sum()
            gsum.backward()
            outs.append((nf.detach(), xyz.grad, feats.grad, [p.grad for p in mod.parameters()],
                         [b.clone() for b in mod.buffers()]))
        finally:
            fused.POINT_SPACE = True
    a, b = outs
    assert _rel(a[0], b[0]) < 1e-5
    if input_grad:
        assert _rel(a[1], b[1]) < 2e-5 and _rel(a[2], b[2]) < 2e-5
    for ga, gb in zip(a[3], b[3]):
        assert _rel(ga, gb) < 2e-5
    for ba, bb in zip(a[4], b[4]):
        assert _rel(ba.float(), bb.float()) < 1e-5


@pytest.mark.parametrize("M,N,K,lda,off", [
    (40000, 64, 132, 135, 3),      # SA1: the cloud's feature columns in place, 128 x 64 tiles
    (70000, 64, 132, 135, 3),      # >= 512 row tiles
    (16384, 128, 128, 128, 0),     # SA2: 64 x 64 tiles
    (70000, 128, 256, 256, 0),     # 128 x 128 tiles
    (1000, 36, 5, 7, 2),           # ragged everything, K < one group
    (333, 256, 259, 259, 0),       # K tail of 3, four column tiles
])
def test_point_gemm_matches_float64(M, N, K, lda, off):
    from scan2cap_amd.pointnet2 import fused
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    buf = torch.randn(M, lda, device="cuda", generator=g)
    A = buf[:, off:off + K]
    Wb = torch.randn(N, K + 3, device="cuda", generator=g) / K ** 0.5
    W = Wb[:, 3:]                                             # a column block, like W[:, 3:]
    P = torch.full((M, N), float("nan"), device="cuda")
    fused._call("s2c_point_gemm", P, M, N, K, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0),
                P.data_ptr(), N)
    want = A.double() @ W.double().t()
    mag = A.double().abs() @ W.double().abs().t()
    assert float(((P.double() - want).abs() / mag.clamp(min=1e-30)).max()) < 1e-6   # an fp32 chain over K
    assert _rel(P, want) < 2e-6
    # the same fp32 chain as the tiled kernel's exact path: bit for bit
    T = torch.empty_like(P)
    prev = fused.set_gemm_split(False)
    try:
        fused._call("s2c_rows_gemm", T, M, N, K, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0),
                    None, None, T.data_ptr(), N, None)
    finally:
        fused.set_gemm_split(prev)
    assert torch.equal(P, T)


@pytest.mark.parametrize("Bn,C,N", [(320000, 132, 64), (160000, 128, 128), (131072 + 17, 132, 64)])
def test_point_gemm_stream_reads_the_cloud_rows_in_place(Bn, C, N):
    """Round 6: the per-point product of a TALL first layer (SA1 at the BASELINE sizes) on the streaming
    kernel -- `s2c_point_gemm_stream`: feature columns of the (.., 3 + C) cloud in place (rows of 3 + C
    floats: neither the stride nor the row starts are 16-byte multiples), bf16x3 products.  Against a
    float64 product (2e-6, the bound of the exact fp32 chain it replaces there) and against
    `s2c_point_gemm`; below the streaming threshold the entry declines (-2) and fused._point_gemm runs
    the exact chain."""
    import ctypes
    from scan2cap_amd import _C
    from scan2cap_amd.pointnet2 import fused
    g = torch.Generator(device="cuda").manual_seed(Bn + C)
    pc = torch.randn(Bn, 3 + C, device="cuda", generator=g)
    f2 = pc[:, 3:]                                   # stride 3 + C, starts 12 bytes into the row
    W = torch.randn(N, 3 + C, device="cuda", generator=g) / (3 + C) ** 0.5
    Wf = W[:, 3:]
    P = torch.full((Bn, N), float("nan"), device="cuda")
    rc = _C.call("s2c_point_gemm_stream", Bn, N, C, f2.data_ptr(), f2.stride(0), Wf.data_ptr(),
                 Wf.stride(0), P.data_ptr(), N, _C.stream_ptr(), allow=(-2,))
    assert rc == 0
    want = f2.double() @ Wf.double().t()
    assert _rel(P, want) < 2e-6
    Q = torch.empty_like(P)
    fused._call("s2c_point_gemm", Q, Bn, N, C, f2.data_ptr(), f2.stride(0), Wf.data_ptr(), Wf.stride(0),
                Q.data_ptr(), N)
    assert _rel(P, Q) < 2e-6
    # via the dispatcher, and a short input stays on the exact chain
    R = torch.full((Bn, N), float("nan"), device="cuda")
    fused._point_gemm(R, f2, Wf, Bn, N, C, 0, 0)
    assert torch.equal(R, P)
    small = 4096
    rc = _C.call("s2c_point_gemm_stream", small, N, C, f2.data_ptr(), f2.stride(0), Wf.data_ptr(),
                 Wf.stride(0), P.data_ptr(), N, _C.stream_ptr(), allow=(-2,))
    assert rc == -2
    S = torch.full((small, N), float("nan"), device="cuda")
    fused._point_gemm(S, f2[:small], Wf, small, N, C, 0, 0)
    assert torch.equal(S, Q[:small])
