"""csrc/s2c_planes.hip (the greedy decoder's bf16x3-plane MFMA GEMMs, models/greedy_fused.py)
against float64 products of the same operands: plane split, generic epilogue (bias, row addend,
ReLU, fp32 + plane outputs, arg-max keys), gathered / two-segment A operands, the GRU-cell epilogue
against torch.nn.GRUCell (models/caption_module.py:254,263), and the greedy feedback through the keys
(caption_module.py:559-566)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _planes_sum(p):
    return p.dense().float().sum(0)


@pytest.fixture(params=[0, 1], ids=["tile128", "tile256"])
def tile(request):
    """Both tile geometries of the kernel (the 256 x 256 one is chosen by grid size otherwise)."""
    from scan2cap_amd.models import greedy_fused as gf
    gf.set_big(request.param)
    yield request.param
    gf.set_big(-1)


def test_planes_split_reconstructs_fp32_and_pads_with_zeros():
    from scan2cap_amd.models import greedy_fused as gf
    torch.manual_seed(0)
    x = torch.randn(37, 300, device="cuda") * torch.logspace(-3, 3, 300, device="cuda")
    xs = torch.zeros(37, 333, device="cuda")
    xs[:, :300] = x                                       # a strided source (row stride 333)
    for src, tiled in ((x, False), (xs[:, :300], False), (x, True), (xs[:, :300], True)):
        p = gf.split(src, rows_out=64, ld=320, tiled=tiled)
        t = p.dense().float()
        assert t.shape == (3, 64, 320)
        if tiled:       # element (r, k) sits where include/s2c_fused.h says
            raw = p.t.view(3, -1).float()
            for r, k in ((0, 0), (9, 5), (9, 13), (36, 299), (31, 17), (24, 8)):
                off = ((r >> 5) * 20 + (k >> 4)) * 512 + \
                    (((r & 31) * 2 + (((k >> 3) & 1) ^ (((r & 31) >> 3) & 1))) << 3) + (k & 7)
                assert float(raw[0, off]) == float(src[r, k].bfloat16().float())
        hi = src.bfloat16().float()
        assert torch.equal(t[0, :37, :300], hi)           # round-to-nearest-even, like torch
        mid = (src - hi).bfloat16().float()
        assert torch.equal(t[1, :37, :300], mid)
        assert torch.equal(t[2, :37, :300], ((src - hi) - mid).bfloat16().float())
        assert float(t[:, 37:].abs().max()) == 0 and float(t[:, :, 300:].abs().max()) == 0
        err = (t.sum(0)[:37, :300].double() - src.double()).abs() / src.abs().double().clamp(min=1e-30)
        assert float(err.max()) < 2.0 ** -22


@pytest.mark.parametrize("M,N,K0,K1,gather", [
    (300, 300, 300, 96, True),        # ragged rows / columns, gathered first segment
    (128, 128, 64, 0, False),         # exactly one tile, one segment
    (1000, 520, 128, 512, False),     # several row and column tiles (XCD map), two segments
    (77, 3500, 512, 0, False),        # the classifier's width
])
def test_planes_gemm_generic_epilogue_matches_float64(M, N, K0, K1, gather, tile):
    from scan2cap_amd.models import greedy_fused as gf
    g = torch.Generator(device="cuda").manual_seed(M + N)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    src_rows = 50 if gather else M
    A0 = rnd(src_rows, K0)
    rowmap = torch.randint(0, src_rows, (M,), device="cuda", generator=g, dtype=torch.int32) \
        if gather else None
    A1 = rnd(M, K1) if K1 else None
    K0p, K1p = gf._up(K0, 32), gf._up(K1, 32)
    W = torch.zeros(N, K0p + K1p, device="cuda")
    W[:, :K0] = rnd(N, K0) / K0 ** 0.5
    if K1:
        W[:, K0p:K0p + K1] = rnd(N, K1) / K1 ** 0.5
    bias, add = rnd(N), rnd(M, N)
    segs = [(gf.split(A0, ld=K0p, tiled=not gather), K0p // 32, rowmap)]
    if K1:
        segs.append((gf.split(A1, ld=K1p), K1p // 32))
    Wp = gf.split(W, rows_out=gf._up(N, 256))
    C = torch.full((M, N), float("nan"), device="cuda")
    P = gf.Planes(M, gf._up(N, 32), "cuda")
    P.t.fill_(float("nan"))
    nct = (N + 127) // 128
    keys = torch.zeros(M, nct, dtype=torch.int64, device="cuda")
    gf.gemm(M, N, segs, Wp, bias=bias, add=add, relu=True, C=C, P=P, amax=keys)
    a0 = A0[rowmap.long()] if gather else A0
    want = a0.double() @ W[:, :K0].double().t() + bias.double() + add.double()
    if K1:
        want = want + A1.double() @ W[:, K0p:K0p + K1].double().t()
    want = want.clamp(min=0)
    assert _rel(C, want) < 2e-6
    ps = _planes_sum(P)
    assert torch.equal(ps[:, :N], C)                      # hi + mid + lo of an fp32 value is exact
    assert float(ps[:, N:].abs().max()) == 0 if P.ld > N else True
    # keys: the first maximum of every row over the columns of each 128-wide tile
    k = keys.cpu().numpy().astype(np.uint64)
    col = (np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64)
    Cc = C.cpu().numpy()
    for ct in range(nct):
        blk = Cc[:, 128 * ct:128 * (ct + 1)]
        assert np.array_equal(col[:, ct], 128 * ct + blk.argmax(1))
    assert np.array_equal(col[np.arange(M), k.argmax(1)], Cc.argmax(1))     # == torch.argmax
    # no bias / add / ReLU, fp32 output only
    C2 = torch.empty(M, N, device="cuda")
    gf.gemm(M, N, segs, Wp, C=C2)
    want2 = a0.double() @ W[:, :K0].double().t()
    if K1:
        want2 = want2 + A1.double() @ W[:, K0p:K0p + K1].double().t()
    assert _rel(C2, want2) < 2e-6


@pytest.mark.parametrize("M,E,H", [(200, 300, 512), (128, 64, 32), (333, 300, 96)])
def test_planes_gemm_gru_epilogue_matches_grucell(M, E, H, tile):
    from scan2cap_amd.models import greedy_fused as gf
    torch.manual_seed(M)
    cell = torch.nn.GRUCell(E, H).cuda()
    x, h = torch.randn(M, E, device="cuda"), torch.randn(M, H, device="cuda")
    Ep = gf._up(E, 32)
    Wg, bg = gf.pack_gru(cell, Ep)
    hn = torch.empty(M, H, device="cuda")
    hp = gf.Planes(M, H, "cuda")
    gf.gemm(M, H, [(gf.split(x, ld=Ep), Ep // 32), (gf.split(h), H // 32)], Wg, bias=bg,
            gru=True, hprev=h, C=hn, P=hp)
    with torch.no_grad():
        want = cell.double()(x.double(), h.double())
    assert _rel(hn, want) < 2e-6
    assert torch.equal(_planes_sum(hp), hn)


def test_greedy_feedback_through_the_argmax_keys(tile):
    """G7 leaves keys, the next G1 gathers the embedding rows of the arg-max tokens."""
    from scan2cap_amd.models import greedy_fused as gf
    torch.manual_seed(3)
    M, V, H, E = 260, 700, 64, 300
    h = torch.randn(M, H, device="cuda")
    Wc = torch.randn(V, H, device="cuda")
    logits = torch.empty(M, V, device="cuda")
    keys = torch.empty(M, (V + 127) // 128, dtype=torch.int64, device="cuda")
    gf.gemm(M, V, [(gf.split(h), H // 32)], gf.split(Wc, rows_out=gf._up(V, 128)), C=logits,
            amax=keys)
    tok = logits.argmax(-1)
    table = torch.randn(V, E, device="cuda")
    Ep = gf._up(E, 32)
    W = torch.zeros(E, Ep, device="cuda")
    W[:, :E] = torch.eye(E, device="cuda")
    out = torch.empty(M, E, device="cuda")
    gf.gemm(M, E, [(gf.split(table, ld=Ep, tiled=False), Ep // 32)],
            gf.split(W, rows_out=gf._up(E, 128)), C=out, tokkeys=keys)
    assert torch.equal(out, table[tok])                   # identity weights: the gathered rows
    # rowdiv: row r reads source row r / 13 (the scene's first word for all of its proposals)
    src = torch.randn(M // 13 + 1, E, device="cuda")
    gf.gemm(M, E, [(gf.split(src, ld=Ep, tiled=False), Ep // 32, 13)],
            gf.split(W, rows_out=gf._up(E, 128)), C=out)
    assert torch.equal(out, src[torch.arange(M, device="cuda") // 13])


def test_planes_gemm_two_outputs_side_by_side(tile):
    """`nsplit`: W rows [0, nsplit) -> C (n1 valid columns; bias, arg-max keys), rows [nsplit, N) -> C2
    (+ row addend): the classifier and map_topdown's h2 block in one pass over h2 (greedy_fused G7)."""
    from scan2cap_amd.models import greedy_fused as gf
    torch.manual_seed(11)
    M, K, n1, n2 = 300, 96, 700, 300
    ns = gf._up(n1, 128)
    A = torch.randn(M, K, device="cuda")
    W1, W2 = torch.randn(n1, K, device="cuda"), torch.randn(n2, K, device="cuda")
    W = torch.zeros(ns + n2, K, device="cuda")
    W[:n1], W[ns:] = W1, W2
    bias, add = torch.randn(n1, device="cuda"), torch.randn(M, n2, device="cuda")
    C = torch.full((M, n1), float("nan"), device="cuda")
    C2 = torch.full((M, n2), float("nan"), device="cuda")
    keys = torch.zeros(M, ns // 128, dtype=torch.int64, device="cuda")
    gf.gemm(M, ns + n2, [(gf.split(A), K // 32)], gf.split(W, rows_out=gf._up(ns + n2, 256)),
            bias=bias, add=add, C=C, amax=keys, split=(ns, n1, C2))
    assert _rel(C, A.double() @ W1.double().t() + bias.double()) < 2e-6
    assert _rel(C2, A.double() @ W2.double().t() + add.double()) < 2e-6
    k = keys.cpu().numpy().astype(np.uint64)
    col = (np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64)
    assert np.array_equal(col[np.arange(M), k.argmax(1)], C.cpu().numpy().argmax(1))


@pytest.mark.parametrize("B,rps,K,H,F", [(2, 48, 48, 512, 128), (3, 13, 300, 256, 64), (16, 512, 512, 512, 128),
                                         (1, 7, 5, 64, 12)])
def test_scene_shared_attention_matches_float64(B, rps, K, H, F):
    """s2c_attn_scene_fwd (num_locals = -1: every row attends over all K objects of its scene)
    against the module's formulation in float64: scores, masked softmax, weighted sum; the bf16x3
    planes of the attended vector re-assemble to it."""
    import ctypes
    from scan2cap_amd import _C
    from scan2cap_amd.models import greedy_fused as gf
    torch.manual_seed(B + K + H)
    R = B * rps
    M = torch.randn(B * K, H, device="cuda") * 0.7
    O = torch.randn(B * K, F, device="cuda")
    q = torch.randn(R, H, device="cuda") * 0.7
    wa = torch.randn(H, device="cuda") * 0.2
    valid = (torch.rand(B, K, device="cuda") < 0.8).float()
    valid[0] = 0 if B > 1 else valid[0]            # a scene without a single valid object: uniform weights
    if B > 1:                                      # huge pre-activations of opposite sign: the exact fallback tile
        M[K + 3, 5], q[rps + 1, 5] = 60.0, -57.5
        M[K + 4, 70 % H], q[rps + 2, 70 % H] = -45.0, 30.0
    alpha = torch.empty(R, K, device="cuda")
    att = torch.empty(R, F, device="cuda")
    Fp = (F + 31) // 32 * 32
    attp = gf.Planes(R, Fp, "cuda", zero=True)
    _C.call("s2c_attn_scene_fwd", R, rps, K, H, F, M.data_ptr(), valid.data_ptr(), O.data_ptr(),
            q.data_ptr(), H, wa.data_ptr(), 0.0, alpha.data_ptr(), att.data_ptr(), F, attp.ptr(),
            attp.pstride, Fp, int(attp.tiled), _C.stream_ptr())
    Md, Od, qd = M.double().view(B, 1, K, H), O.double().view(B, K, F), q.double().view(B, rps, 1, H)
    s = (torch.tanh(Md + qd) * wa.double()).sum(-1)                       # (B, rps, K)
    s = s.masked_fill(valid.view(B, 1, K) == 0, -1e30)
    a = torch.softmax(s, -1)
    want = a @ Od                                                            # (B, rps, F)
    assert (alpha.double().view(B, rps, K) - a).abs().max() < 2e-6
    assert ((att.double().view(B, rps, F) - want).abs().max() / want.abs().max()) < 2e-6
    planes = attp.dense().float().sum(0)[:, :F]
    assert (planes - att).abs().max() < 2e-6 * max(1.0, att.abs().max().item())
