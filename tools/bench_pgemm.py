"""s2c_point_gemm against the tiled exact-fp32 and bf16x3 kernels at the cfg3 step's point products."""
import sys
import torch
sys.path.insert(0, ".")
from scan2cap_amd.pointnet2 import fused

def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for (M, N, K, lda, off) in ((320000, 64, 132, 135, 3), (16384, 128, 128, 128, 0), (8192, 128, 256, 256, 0),
                            (4096, 128, 256, 256, 0), (1280000, 64, 132, 135, 3)):
    buf = torch.randn(M, lda, device="cuda")
    A = buf[:, off:off + K]
    Wb = torch.randn(N, K + 3, device="cuda")
    W = Wb[:, 3:]
    P = torch.empty(M, N, device="cuda")
    new = lambda: fused._call("s2c_point_gemm", P, M, N, K, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), P.data_ptr(), N)
    old = lambda: fused._call("s2c_rows_gemm", P, M, N, K, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), None, None, P.data_ptr(), N, None)
    t_new = timeit(new)
    fused.set_gemm_split(False); t_exact = timeit(old)
    fused.set_gemm_split(True); t_x3 = timeit(old)
    gb = 4 * (M * K + M * N) / 1e3
    print("(%8d,%4d,%4d) point_gemm %7.1f us (%.2f TB/s, %5.1f TF) | tiled exact %7.1f | tiled bf16x3 %7.1f" % (
        M, N, K, t_new, gb / t_new / 1e3, 2.0 * M * N * K / t_new / 1e6, t_exact, t_x3))
