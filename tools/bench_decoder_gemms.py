"""The decoder's recurrence-free GEMMs (cfg3: R = 8 rows, T = 30 steps, K = 10 objects), each
timed alone inside a captured hipGraph of 20 back-to-back calls: library (torch.mm) against
s2c_weight_grad for the dW = X^T Y shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scan2cap_amd.pointnet2 import fused

dev = torch.device("cuda:0")
TR = 240
DW = [("dW_cls", TR, 3500, 512), ("dW_td.w", TR, 300, 300), ("dW_td.h", TR, 300, 512),
      ("dW_td.tf", 8, 300, 128), ("dW_ih1", TR, 1536, 300), ("dW_hh1", TR, 1536, 512),
      ("dW_h", TR, 512, 512), ("dW_lang.a", TR, 300, 128), ("dW_lang.h", TR, 300, 512),
      ("dW_ih2", TR, 1536, 300), ("dW_hh2", TR, 1536, 512), ("dW_f", 80, 512, 128)]
DA = [("dH2", TR, 3500, 512), ("dtf", 8, 300, 128), ("dO2", 80, 512, 128), ("logits", TR, 512, 3500),
      ("Pw", TR, 300, 300), ("M", 80, 128, 512)]


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

tot_l = tot_h = 0.0
for name, M, Co, Ci in DW:
    dY = torch.randn(M, Co, device=dev); A = torch.randn(M, Ci, device=dev)
    out = torch.empty(Co, Ci, device=dev)
    tl = timed(lambda: torch.mm(dY.t(), A, out=out))
    th = timed(lambda: fused.weight_grad_kernel(dY, A))
    ref = torch.mm(dY.double().t(), A.double())
    err = float((fused.weight_grad_kernel(dY, A).double() - ref).abs().max() / ref.abs().max())
    tot_l += tl; tot_h += th
    print("%-10s (%4d,%4d,%4d)  library %6.1f us   s2c_weight_grad %6.1f us   err %.1e" % (name, M, Co, Ci, tl, th, err))
print("dW total: library %.1f us, hand %.1f us" % (tot_l, tot_h))
for name, M, K, N in DA:
    X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    print("%-10s (%4d,%4d)x(%4d,%4d)  library %6.1f us" % (name, M, K, K, N, timed(lambda: torch.mm(X, W, out=out))))
