"""Tall weight gradients: csrc/s2c_dw32.hip (fp32 MFMA, row-major operands straight from memory) against
the split-K library bmm + partial sums of fused._weight_grad, at the cfg3 step's shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan2cap_amd.pointnet2 import fused  # noqa: E402


def timed(fn, reps=10):
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


for M, C, K in [(1048576, 64, 64), (262144, 128, 128), (262144, 256, 128), (262144, 128, 131),
                (65536, 256, 128), (65536, 128, 259), (32768, 128, 259), (320000, 64, 132)]:
    dY, X = torch.randn(M, C, device="cuda"), torch.randn(M, K, device="cuda")

    def run(on):
        fused.USE_DW32 = on
        pend = []
        fused._weight_grad(dY, X, pend)
        fused.flush_partial_sums(pend)
    t1, t0 = timed(lambda: run(True)), timed(lambda: run(False))
    gb = 4.0 * M * (C + K) / 1e3
    print("(%8d,%4d,%4d)  dw32 %7.1f us (%.2f TB/s, %5.1f TF)   library %7.1f us" % (
        M, C, K, t1, gb / t1 / 1e3, 2.0 * M * C * K / t1 / 1e6, t0))
