"""Time the planes GEMMs (csrc/s2c_planes.hip) at the greedy decoder's shapes against torch's
fp32 GEMM of the same product: us per launch, fp32-equivalent TFLOP/s, fraction of the dense bf16
MFMA roof (6 plane products per fp32 product).  python tools/bench_planes.py [R ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan2cap_amd.models import greedy_fused as gf  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    Rs = [int(a) for a in sys.argv[1:]] or [2048, 8192]
    dbg = int(os.environ.get("S2C_PLANES_DBG", "0"))     # bit 0: no MFMA, bit 1: no DMA (timing only)
    gf.DEBUG = dbg
    E, H, F, V = 300, 512, 128, 3500
    Ep = 320
    dev = "cuda"
    for R in Rs:
        print("R = %d" % R)
        rnd = lambda *s: torch.randn(*s, device=dev)
        cell = torch.nn.GRUCell(E, H).to(dev)
        Wg, bg = gf.pack_gru(cell, Ep)
        x, h = rnd(R, E), rnd(R, H)
        xp, hp = gf.split(x, ld=Ep), gf.split(h)
        hn, hnp = torch.empty(R, H, device=dev), gf.Planes(R, H, dev)
        shapes = []
        # (name, N, [K segments], fn, torch fn)
        W1 = gf.split(rnd(E, Ep + H), rows_out=384)
        add = rnd(R, E)
        outp = gf.Planes(R, Ep, dev)
        shapes.append(("G1 map_topdown  N=300 K=320+512", E, Ep + H,
                       lambda: gf.gemm(R, E, [(xp, 10), (hp, 16)], W1, add=add, relu=True, P=outp)))
        shapes.append(("G2 GRU cell     N=3x512 K=320|512", 3 * H, (Ep + H) * 2 / 3.0,
                       lambda: gf.gemm(R, H, [(xp, 10), (hp, 16)], Wg, bias=bg, gru=True, hprev=h,
                                       C=hn, P=hnp)))
        Wq = gf.split(rnd(H, H), rows_out=512)
        q = torch.empty(R, H, device=dev)
        shapes.append(("G3 map_hidd     N=512 K=512", H, H,
                       lambda: gf.gemm(R, H, [(hp, 16)], Wq, C=q)))
        Wc = gf.split(rnd(V, H), rows_out=3584)
        bc = rnd(V)
        logits = torch.empty(R, V, device=dev)
        keys = torch.empty(R, 28, dtype=torch.int64, device=dev)
        shapes.append(("G7 classifier   N=3500 K=512", V, H,
                       lambda: gf.gemm(R, V, [(hp, 16)], Wc, bias=bc, C=logits, amax=keys)))
        tot = 0.0
        for name, N, K, fn in shapes:
            us = timed(fn)
            fl = 2.0 * R * N * K
            tot += us * (2 if name.startswith(("G1", "G2")) else 1)
            print("  %-36s %8.1f us  %6.1f TF fp32-equivalent  %.3f of the bf16 MFMA roof"
                  % (name, us, fl / us / 1e6, 6 * fl / us / 1e6 / 2500.0))
        a, b = rnd(R, H), rnd(V, H)
        us = timed(lambda: torch.mm(a, b.t()))
        print("  torch.mm fp32 (R,512)x(512,3500): %.1f us = %.1f TF" % (us, 2.0 * R * V * H / us / 1e6))
        a, b = rnd(R, Ep + H), rnd(3 * H, Ep + H)
        us = timed(lambda: torch.mm(a, b.t()))
        print("  torch.mm fp32 (R,832)x(832,1536): %.1f us" % us)
        print("  GEMMs of one token (2 x G1, 2 x G2, G3, G7): %.0f us" % tot)


if __name__ == "__main__":
    main()
