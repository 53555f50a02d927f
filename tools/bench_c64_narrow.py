"""rows_gemm_c64_kernel: 128 x 128 vs 128 x 32 workgroup tiles on the mid layers of the cfg3 step (us,
inside a replayed hipGraph)."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from scan2cap_amd import _C
from scan2cap_amd.pointnet2 import fused

lib = _C.load()
lib.s2c_gemm_set_c64_narrow.argtypes = [ctypes.c_int]; lib.s2c_gemm_set_c64_narrow.restype = ctypes.c_int
lib.s2c_rows_gemm_blocks.argtypes = [ctypes.c_longlong, ctypes.c_int]; lib.s2c_rows_gemm_blocks.restype = ctypes.c_int

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

for (M, N, K) in ((8192, 256, 256), (16384, 128, 128), (16384, 256, 256), (20480, 128, 128), (20480, 256, 128),
                  (32768, 128, 128), (32768, 256, 128), (65536, 128, 128), (4096, 256, 256)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    part = torch.empty(lib.s2c_rows_gemm_blocks(M, N) * 2 * N, device="cuda")
    f = lambda: fused._call("s2c_rows_gemm", Y, M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, None, Y.data_ptr(), N, part.data_ptr())
    t = {}
    for v in (1, 0):
        lib.s2c_gemm_set_c64_narrow(v)
        t[v] = graph_time(f)
    lib.s2c_gemm_set_c64_narrow(1)
    print("(%6d, %3d, %3d) tiles128 %4d: narrow %6.1f us | wide %6.1f us" % (M, N, K, ((M + 127) // 128) * ((N + 127) // 128), t[1], t[0]))
