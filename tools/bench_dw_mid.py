"""dW = dY^T A at the mid-size layer shapes (M = 16k..262k rows): split-K library bmm + partial
sums vs the hand-written slab kernel (s2c_dw.hip) + partial sums -- where is the crossover?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import fused
from tools.bench_ops import timeit

def bench(M, C, N):
    r = lambda *s: torch.randn(*s, device="cuda")
    dY, A = r(M, C), r(M, N)
    pend = []
    def dw_hand():
        fused._weight_grad_partials(dY, A, pend); fused.flush_partial_sums(pend)
    def dw_lib():
        fused._weight_grad(dY, A, pend); fused.flush_partial_sums(pend)
    old = fused._hand_dw_pays
    fused._hand_dw_pays = lambda *a: False
    a = fused._weight_grad(dY, A, pend); fused.flush_partial_sums(pend)
    t_l = timeit(dw_lib)
    fused._hand_dw_pays = old
    b = fused._weight_grad_partials(dY, A, pend); fused.flush_partial_sums(pend)
    t_h = timeit(dw_hand)
    ref = dY.double().t() @ A.double()
    e = lambda x: float((x.double() - ref).abs().max() / ref.abs().max())
    print("M=%7d C=%3d N=%3d | dW lib %6.1f us  hand %6.1f us | err lib %.1e hand %.1e" % (M, C, N, t_l, t_h, e(a), e(b)))

for shp in ((16384, 64, 3), (20480, 128, 128), (20480, 128, 256), (20480, 256, 128), (32768, 128, 128),
            (32768, 128, 259), (32768, 256, 128), (65536, 128, 128), (65536, 128, 259), (65536, 256, 128),
            (262144, 128, 128), (262144, 128, 131), (262144, 256, 128), (320000, 64, 132), (320000, 64, 3)):
    bench(*shp)
