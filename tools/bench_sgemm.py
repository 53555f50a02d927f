"""The small products of the cfg3 step (tools/lib_gemm_census.py): csrc/s2c_sgemm.hip against
torch.mm / torch.addmm, kernel durations from a rocprofv3 kernel trace (tools/trace_blocks.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan2cap_amd.pointnet2 import fused  # noqa: E402

for M, K, N, tr in [(8192, 256, 256, False), (20480, 128, 128, False), (20480, 128, 256, False),
                    (8192, 256, 512, False), (8192, 259, 256, False), (4096, 256, 512, False),
                    (4096, 256, 256, False), (2048, 128, 128, False), (2048, 97, 128, False),
                    (8192, 256, 256, True)]:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") if tr else torch.randn(K, N, device="cuda")
    bias = torch.randn(N, device="cuda") if tr else None
    for _ in range(23):
        fused.small_gemm(A, B, tr, bias)
    torch.cuda.synchronize()
    for _ in range(23):
        if tr:
            torch.addmm(bias, A, B.t())
        else:
            torch.mm(A, B)
    torch.cuda.synchronize()
    print("done", M, K, N, tr)
