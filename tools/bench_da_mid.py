"""dX = dY W at the mid-size layer shapes: torch.mm (hipBLASLt) vs the hand-written rows GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import fused
from tools.bench_ops import timeit

def bench(M, C, N):
    dY, W = torch.randn(M, C, device="cuda"), torch.randn(C, N, device="cuda") * 0.1
    a = torch.mm(dY, W)
    b = fused._input_grad_gemm(dY, W)
    ref = dY.double() @ W.double()
    e = lambda x: float((x.double() - ref).abs().max() / ref.abs().max())
    t_l = timeit(lambda: torch.mm(dY, W))
    t_h = timeit(lambda: fused._input_grad_gemm(dY, W))
    Wt = W.t().contiguous()
    print("M=%7d C=%3d N=%3d | dX lib %6.1f us  hand %6.1f us (incl. W^T copy) | err lib %.1e hand %.1e" % (
        M, C, N, t_l, t_h, e(a), e(b)))

for shp in ((2048, 97, 128), (2048, 128, 128), (4096, 256, 256), (4096, 256, 512), (8192, 256, 256), (8192, 256, 512),
            (8192, 259, 256), (20480, 128, 128), (20480, 128, 256), (32768, 128, 128), (32768, 128, 259),
            (32768, 256, 128), (65536, 128, 259), (65536, 256, 128), (262144, 128, 131), (262144, 256, 128)):
    bench(*shp)
