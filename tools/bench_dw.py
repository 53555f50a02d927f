"""dW = dY^T A at the layer shapes of the cfg3 train step: the hand-written kernel
(csrc/s2c_dw.hip) against the previous path (split-K strided-batched library GEMM + partial
sum) -- time per call and fraction of the HBM roof (bytes = 4 M (Cout + Cin))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import fused

SHAPES = [(1 << 20, 64, 64), (1 << 20, 128, 64), (262144, 128, 131), (262144, 128, 128),
          (262144, 256, 128), (65536, 128, 259), (65536, 128, 128), (65536, 256, 128),
          (32768, 128, 259), (32768, 256, 128), (8192, 256, 512), (8192, 256, 256),
          (8192, 259, 256), (4096, 256, 512), (2048, 128, 128), (2048, 97, 128),
          (20480, 128, 256), (20480, 128, 128)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    tot_k = tot_l = 0.0
    for M, Cout, Cin in SHAPES:
        dY = torch.randn(M, Cout, device="cuda") * 0.1
        A = torch.randn(M, Cin, device="cuda")
        fused.USE_DW_KERNEL = True
        tk = timeit(lambda: fused._weight_grad(dY, A))
        fused.USE_DW_KERNEL = False
        tl = timeit(lambda: fused._weight_grad(dY, A))
        # main loop only: partial tiles per slab, no cross-workgroup reduction
        from scan2cap_amd import _C
        wb, cb = fused._dw_sizes(M, Cout, Cin)
        work = torch.empty(max(wb // 4, 1), device="cuda")
        dW = torch.empty(Cout, Cin, device="cuda")
        tp = timeit(lambda: _C.call("s2c_weight_grad", M, Cout, Cin, dY.data_ptr(), dY.stride(0),
                                    A.data_ptr(), A.stride(0), dW.data_ptr(), Cin,
                                    work.data_ptr(), None, _C.stream_ptr()))
        gb = 4.0 * M * (Cout + Cin) / 1e9
        tot_k += tk
        tot_l += tl
        print("M=%8d Cout=%3d Cin=%3d  kernel %7.1f us (%.2f of 8 TB/s)   library %7.1f us"
              % (M, Cout, Cin, tk, gb / (tk * 1e-6) / 8000, tl), "  partials only %7.1f us" % tp)
    print("sum: kernel %.0f us, library %.0f us" % (tot_k, tot_l))


if __name__ == "__main__":
    main()
