"""Split-K choice of the library weight-gradient path (fused._weight_grad): time of
bmm + partial sum for S = 1 .. 512 slabs at the layer shapes of the cfg3 train step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SHAPES = [(1 << 20, 64, 64), (1 << 20, 128, 64), (262144, 128, 131), (262144, 128, 128),
          (262144, 256, 128), (65536, 128, 259), (65536, 128, 128), (65536, 256, 128),
          (32768, 128, 259), (32768, 128, 128), (32768, 256, 128), (8192, 256, 512),
          (8192, 256, 256), (8192, 259, 256), (4096, 256, 512), (2048, 128, 128),
          (20480, 128, 256), (20480, 128, 128)]


def timeit(fn, n=30):
    g = torch.cuda.CUDAGraph()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for M, Cout, Cin in SHAPES:
    dY = torch.randn(M, Cout, device="cuda") * 0.1
    A = torch.randn(M, Cin, device="cuda")
    res = []
    S = 1
    while S <= 512 and M % S == 0 and M // S >= 512:
        if S == 1:
            t = timeit(lambda: torch.mm(dY.t(), A))
        else:
            t = timeit(lambda: torch.bmm(dY.view(S, M // S, -1).transpose(1, 2),
                                         A.view(S, M // S, -1)).sum(0))
        res.append((S, t))
        S *= 2
    best = min(res, key=lambda r: r[1])
    cur = 1
    while cur < 512 and M % (2 * cur) == 0 and M // (2 * cur) >= 1024:
        cur *= 2
    tcur = dict(res).get(cur)
    print("M=%8d %3dx%3d  current S=%3d %6.1f us   best S=%3d %6.1f us   all: %s"
          % (M, Cout, Cin, cur, tcur, best[0], best[1],
             " ".join("%d:%.0f" % r for r in res)))
