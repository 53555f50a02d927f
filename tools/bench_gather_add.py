"""s2c_sa_gather_add at the cfg3 step's five stages (us, TB/s of the Y write + P reads)."""
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

for (B, n, m, ns, N) in ((8, 40000, 2048, 64, 64), (8, 2048, 1024, 32, 128), (8, 1024, 512, 16, 128),
                         (8, 512, 256, 16, 128), (8, 1024, 256, 16, 128)):
    xyz = torch.rand(B, n, 3, device="cuda") * 6
    new_xyz = xyz[:, :m].contiguous()
    # neighbours of a centre are near each other in memory order, like a ball query's
    idx = ((torch.arange(m, device="cuda").view(1, m, 1) * (n // m) + torch.randint(0, 64, (B, m, ns), device="cuda")) % n).int()
    P = torch.randn(B * n, N, device="cuda")
    W = torch.randn(N, 3 + 8, device="cuda")
    rows = B * m * ns
    Y = torch.empty(rows, N, device="cuda")
    part = torch.empty(fused._gather_add_blocks(rows) * 2 * N, device="cuda")
    f = lambda: fused._call("s2c_sa_gather_add", Y, B, n, m, ns, N, 0.2, 1, xyz.data_ptr(), new_xyz.data_ptr(),
                            P.data_ptr(), idx.data_ptr(), W.data_ptr(), W.stride(0), Y.data_ptr(), part.data_ptr())
    t = timeit(f)
    print("(B %d n %5d m %4d ns %2d N %3d) rows %7d: %6.1f us, Y write %.2f TB/s" % (B, n, m, ns, N, rows, t, 4.0 * rows * N / t / 1e6))
