"""csrc/s2c_mgemm.hip (many small fp32 GEMMs in one launch: the teacher-forced decoder's hoisted
products, models/decoder_fused.py) against float64 products: transposed / column-block / two-level
row-order operands, bias, accumulate, split-K, ragged sizes, more jobs than one launch takes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def test_mgemm_jobs_match_float64():
    from scan2cap_amd import mgemm as mg
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    jobs, want = [], []
    # plain, ragged
    A, B = rnd(70, 33), rnd(33, 130)
    C0 = torch.empty(70, 130, device="cuda")
    jobs.append(mg.mm(A, B, C0)); want.append((C0, A.double() @ B.double()))
    # A transposed view, B = column block of a weight used transposed, bias, C = column block
    G, X, W = rnd(240, 300), rnd(240, 512), rnd(300, 940)
    big = torch.zeros(300, 940, device="cuda")
    jobs.append(mg.mm(G.t(), X, big[:, 300:812])); want.append((big[:, 300:812], G.double().t() @ X.double()))
    bias = rnd(300)
    C2 = torch.empty(8, 300, device="cuda")
    tf = rnd(8, 128)
    jobs.append(mg.mm(tf, W[:, 812:].t(), C2, bias=bias))
    want.append((C2, tf.double() @ W[:, 812:].double().t() + bias.double()))
    # accumulate
    C3 = rnd(64, 64)
    base = C3.clone()
    A3, B3 = rnd(64, 17), rnd(17, 64)
    jobs.append(mg.mm(A3, B3, C3, accumulate=True)); want.append((C3, base.double() + A3.double() @ B3.double()))
    # two-level row order: rows (r, t) of the output read a (T, R, H) tensor's rows (t, r)
    T, R, H, V = 7, 5, 96, 150
    H2, Wc = rnd(T, R, H), rnd(V, H)
    C4 = torch.empty(R, T, V, device="cuda")
    jobs.append(mg.Job(R * T, V, H, H2, mg.ax(R * H, div=T, hi=H), mg.ax(1), Wc, mg.ax(1), mg.ax(H),
                       C4, mg.ax(V)))
    want.append((C4, H2.permute(1, 0, 2).double() @ Wc.double().t()))
    # split-K into a zeroed C addressed through the two-level map
    dl = rnd(R * T, 3500)
    Wbig = rnd(3500, H)
    C5 = torch.zeros(T, R, H, device="cuda")
    jobs.append(mg.Job(R * T, H, 3500, dl, mg.ax(3500), mg.ax(1), Wbig, mg.ax(H), mg.ax(1),
                       C5, mg.ax(R * H, div=T, hi=H), ksplit=16))
    want.append((C5, (dl.double() @ Wbig.double()).view(R, T, H).permute(1, 0, 2)))
    # more jobs than one launch holds
    extra = []
    for i in range(40):
        a, b = rnd(9 + i, 5 + i), rnd(5 + i, 11)
        c = torch.empty(9 + i, 11, device="cuda")
        extra.append((a, b, c))
        jobs.append(mg.mm(a, b, c)); want.append((c, a.double() @ b.double()))
    mg.launch(jobs)
    for i, (got, ref) in enumerate(want):
        assert _rel(got, ref) < 3e-6, i
    assert float(big[:, :300].abs().max()) == 0 and float(big[:, 812:].abs().max()) == 0
