"""Tall weight gradients: the streaming kernel (csrc/s2c_dwstream.hip) against the split-K library bmm
+ partial sums of fused._weight_grad, at the cfg3 step's shapes (tools/lib_gemm_census.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan2cap_amd.pointnet2 import fused  # noqa: E402


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


cloud = torch.randn(8 * 40000, 135, device="cuda")
cases = [("1M 64x64", 1 << 20, 64, 64, None), ("gram 1M 64", 1 << 20, 64, 64, "same"),
         ("SA2-L3", 262144, 256, 128, None), ("SA2-L2", 262144, 128, 128, None),
         ("SA3", 65536, 256, 128, None), ("pts 64x132", 320000, 64, 132, "feats"),
         ("pts 64x135", 320000, 64, 135, "cloud"), ("pts 64x3", 320000, 64, 3, None),
         ("SA2-L1 128x131", 262144, 128, 131, None)]
for name, M, C, N, kind in cases:
    dY = torch.randn(M, C, device="cuda")
    if kind == "same":
        X = dY
    elif kind == "feats":
        X = cloud[:, 3:]
    elif kind == "cloud":
        X = cloud
    else:
        X = torch.randn(M, N, device="cuda")

    def run(on):
        fused.DW_STREAM = on
        pend = []
        fused._weight_grad(dY, X if on or X.is_contiguous() or kind == "same" else X.contiguous(), pend)
        fused.flush_partial_sums(pend)
    Xc = X if X.is_contiguous() else X.contiguous()

    def lib():
        fused.DW_STREAM = False
        pend = []
        fused._weight_grad(dY, Xc, pend)
        fused.flush_partial_sums(pend)
    t1, t0 = timed(lambda: run(True)), timed(lib)
    fused.DW_STREAM = True
    gb = 4.0 * M * ((C if kind == "same" else C + N)) / 1e3
    print("%-16s (%8d,%4d,%4d)  stream %7.1f us (%.2f TB/s)   library %7.1f us" % (
        name, M, C, N, t1, gb / t1 / 1e3, t0))
