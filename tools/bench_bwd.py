"""Backward of one BN+ReLU+linear layer at the SA1 / SA2 shapes: separate passes
(s2c_bn_relu_bwd + torch.mm + split-K library dW) vs the hand-written path
(s2c_bn_relu_bwd_stats + s2c_bn_bwd_gemm + s2c_weight_grad partials)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import fused
from tools.bench_ops import timeit

def bench(M, C, N):
    r = lambda *s: torch.randn(*s, device="cuda")
    Y, dA, W, A = r(M, C), r(M, C), r(C, N) * 0.1, r(M, N)
    gamma = torch.rand(C, device="cuda") + 0.5
    mean, invstd = Y.mean(0), 1.0 / torch.sqrt(Y.var(0, unbiased=False) + 1e-5)
    scale = gamma * invstd; shift = -mean * scale
    nb = fused._stat_blocks(M)
    partial = torch.empty(nb * 2 * max(C, 256), device="cuda")
    coef, dg, db = torch.empty(3 * C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dY = torch.empty_like(Y); dX = torch.empty(M, N, device="cuda")
    Wt = W.t().contiguous()
    common = (dA.data_ptr(), Y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr())
    t_stats = timeit(lambda: fused._call("s2c_bn_relu_bwd_stats", Y, M, C, *common, gamma.data_ptr(), 1, 0, partial.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr()))
    t_full = timeit(lambda: fused._call("s2c_bn_relu_bwd", Y, M, C, *common, gamma.data_ptr(), 1, 0, partial.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(), dY.data_ptr()))
    t_mm = timeit(lambda: torch.mm(dY, W))
    t_fused = timeit(lambda: fused._call("s2c_bn_bwd_gemm", Y, M, C, N, *common, coef.data_ptr(), 1, Wt.data_ptr(), Wt.stride(0), dY.data_ptr(), dX.data_ptr(), N))
    t_hand = timeit(lambda: fused._input_grad_gemm(dY, W))
    pend = []
    def dw_hand():
        fused._weight_grad_partials(dY, A, pend); fused.flush_partial_sums(pend)
    fused._hand_dw_pays = lambda *a: False
    def dw_lib():
        fused._weight_grad(dY, A, pend); fused.flush_partial_sums(pend)
    t_dwh, t_dwl = timeit(dw_hand), timeit(dw_lib)
    GB = lambda units: 4e-9 * M * units
    print("M=%7d C=%3d N=%3d | apply-only %5.0f us | library dA %5.0f (%.2f TB/s) hand dA %5.0f | apply+lib %5.0f  fused %5.0f (%.2f TB/s) | dW lib %5.0f hand %5.0f (%.2f TB/s)" % (
        M, C, N, t_full - t_stats, t_mm, GB(C + N) / t_mm * 1e3, t_hand, t_full - t_stats + t_mm, t_fused, GB(3 * C + N) / t_fused * 1e3, t_dwl, t_dwh, GB(C + N) / t_dwh * 1e3))

for shp in ((1048576, 64, 64), (1048576, 128, 64), (262144, 128, 128), (262144, 256, 128), (262144, 128, 131),
            (65536, 256, 128), (65536, 128, 259), (8192, 256, 256)):
    bench(*shp)
