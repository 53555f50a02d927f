"""Kernel time of the fused 64 -> 64 layer backward (csrc/s2c_bnbwd_fused.hip) beside the launches it
replaces (s2c_bn_bwd_gemm_next_stats + s2c_weight_grad_stream), torch events over 20 launches.
    python tools/bench_bnbwd_fused.py [M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan2cap_amd.pointnet2 import fused
from tests.test_bnbwd_fused_gpu import _inputs, _run

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
d = _inputs(M, 1, 1, 1)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


parts = fused._bwd_dx_dw_parts(M)
dX = torch.empty(M, 64, device="cuda"); wpart = torch.empty(parts, 64, 64, device="cuda")
npart = torch.empty(parts, 128, device="cuda")


def new():
    fused._call("s2c_bn_bwd_dx_dw64", dX, M, d["dA"].data_ptr(), d["Y"].data_ptr(),
                d["scale"].data_ptr(), d["shift"].data_ptr(), d["mean"].data_ptr(),
                d["invstd"].data_ptr(), d["coef"].data_ptr(), 1, d["W"].data_ptr(), 64, dX.data_ptr(),
                d["nY"].data_ptr(), d["nscale"].data_ptr(), d["nshift"].data_ptr(), d["nmean"].data_ptr(),
                d["ninvstd"].data_ptr(), 1, wpart.data_ptr(), npart.data_ptr())


Wt = d["W"].t().contiguous(); dY = torch.empty(M, 64, device="cuda")
nbg = fused._gemm_blocks(M, 64); np2 = torch.empty(nbg * 128, device="cuda")
act = torch.relu(d["nY"] * d["nscale"] + d["nshift"])


def old_gemm():
    fused._call("s2c_bn_bwd_gemm_next_stats", dX, M, 64, 64, d["dA"].data_ptr(), d["Y"].data_ptr(),
                d["scale"].data_ptr(), d["shift"].data_ptr(), d["mean"].data_ptr(),
                d["invstd"].data_ptr(), d["coef"].data_ptr(), 1, Wt.data_ptr(), 64, dY.data_ptr(),
                dX.data_ptr(), 64, d["nY"].data_ptr(), d["nscale"].data_ptr(), d["nshift"].data_ptr(),
                d["nmean"].data_ptr(), d["ninvstd"].data_ptr(), 1, np2.data_ptr())


def old_dw():
    pend = []
    fused._weight_grad_stream(dY, act, pend)


t_new, t_g, t_w = timed(new), timed(old_gemm), timed(old_dw)
print("M=%d  fused %.1f us (%.2f TB/s of 1024 B/row)   bn_bwd_gemm_next_stats %.1f + weight_grad_stream %.1f us"
      % (M, t_new, M * 1024 / t_new / 1e6, t_g, t_w))
