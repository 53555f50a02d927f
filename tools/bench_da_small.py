"""dX = dY W below 32768 rows: hipBLASLt (torch.mm) vs s2c_point_gemm on W^T (incl. the transpose copy)
vs the tiled bf16x3 kernel (_input_grad_gemm), inside a replayed hipGraph (us per call)."""
import sys
import torch
sys.path.insert(0, ".")
from scan2cap_amd.pointnet2 import fused

def graph_time(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * n) * 1e3

for (M, Cout, Cin) in ((8192, 256, 256), (20480, 128, 128), (20480, 128, 256), (8192, 256, 512), (8192, 259, 256),
                       (4096, 256, 512), (4096, 256, 256), (2048, 128, 128), (2048, 97, 128)):
    dY = torch.randn(M, Cout, device="cuda")
    W = torch.randn(Cout, Cin, device="cuda")
    out = torch.empty(M, Cin, device="cuda")
    lib = lambda: torch.mm(dY, W, out=out)
    def pg():
        Wt = W.t().contiguous()
        fused._call("s2c_point_gemm", out, M, Cin, Cout, dY.data_ptr(), dY.stride(0), Wt.data_ptr(), Wt.stride(0),
                    out.data_ptr(), Cin)
    x3 = lambda: fused._input_grad_gemm(dY, W)
    print("(%6d, %3d -> %3d)  library %6.1f us | point_gemm %6.1f | tiled bf16x3 %6.1f" % (M, Cout, Cin, graph_time(lib), graph_time(pg), graph_time(x3)))
